"""scratch (round 2): does the ORDER of the (query, node) pairs matter to cz_distance_batch at 10M base rows (30 GB)?
random pairs / pairs sorted by node id / pairs sorted by node >> 13 (25 MB buckets), and the same over 1M rows"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.hnsw import distance_batch_device

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
stream = torch.cuda.current_stream().cuda_stream
q = Bn.gen_vectors(torch, 1024, 768, "lowrank", 43, dev)
P = 1 << 22
for n in (10_000_000, 1_000_000):
    x = Bn.gen_vectors(torch, n, 768, "lowrank", 42, dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    qi = torch.randint(0, 1024, (P,), generator=g, device=dev, dtype=torch.int32)
    ni = torch.randint(0, n, (P,), generator=g, device=dev, dtype=torch.int32)
    forms = {"random": torch.arange(P, device=dev),
             "sorted by node": torch.argsort(ni.to(torch.int64)),
             "sorted by node>>13": torch.argsort((ni >> 13).to(torch.int64), stable=True),
             "sorted by node>>16": torch.argsort((ni >> 16).to(torch.int64), stable=True)}
    out = torch.empty(P, dtype=torch.float64, device=dev)
    for name, perm in forms.items():
        pairs = torch.stack([qi[perm], ni[perm]], 1).contiguous()
        for _ in range(2):
            distance_batch_device("Cosine", x, q, pairs, out, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            distance_batch_device("Cosine", x, q, pairs, out, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"n={n} pairs {name:20s}: {ms:.3f} ms  {P * 768 * 4 / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    del x
    torch.cuda.empty_cache()

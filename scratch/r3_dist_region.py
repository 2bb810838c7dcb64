"""scratch (round 3): cz_distance_batch on the 10M x 768 table with and without the region ordering of the pairs
(CZ_PAIRS_REGION, CZ_PAIRS_REGION_SHIFT): whole-call time by HIP events, outputs compared bit for bit."""
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
    pairs = torch.stack([torch.randint(0, 1024, (P,), generator=g, device=dev, dtype=torch.int32),
                         torch.randint(0, n, (P,), generator=g, device=dev, dtype=torch.int32)], 1).contiguous()
    ref = None
    for name, env in (("caller order", {"CZ_PAIRS_REGION": "0"}), ("default", {}), ("region >>13", {"CZ_PAIRS_REGION": "1", "CZ_PAIRS_REGION_SHIFT": "13"}),
                      ("region >>16", {"CZ_PAIRS_REGION": "1", "CZ_PAIRS_REGION_SHIFT": "16"}),
                      ("region >>11", {"CZ_PAIRS_REGION": "1", "CZ_PAIRS_REGION_SHIFT": "11"}),
                      ("caller order again", {"CZ_PAIRS_REGION": "0"})):
        for k in ("CZ_PAIRS_REGION", "CZ_PAIRS_REGION_SHIFT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        out = torch.full((P,), float("nan"), dtype=torch.float64, device=dev)
        for _ in range(2):
            distance_batch_device("Cosine", x, q, pairs, out, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            distance_batch_device("Cosine", x, q, pairs, out, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        same = "" if ref is None else f"  identical to caller order: {bool(torch.equal(out, ref))}"
        if ref is None:
            ref = out.clone()
        print(f"n={n} {name:20s}: {ms:.3f} ms whole call  {P * 768 * 4 / ms / 1e6 / 8000:.3f} of 8 TB/s{same}", flush=True)
    del x
    torch.cuda.empty_cache()

"""scratch (round 5): does the level a landing gives show in the translation counters?  One quick 10M build, then handles created again
and again WITHOUT the placement trial (CZ_TABLE_SETTLE=0), each timed with 4 + 24 launches.  Run under
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum --kernel-include-regex hnsw_knn_kernel
the per-dispatch counters line up with the printed fractions (28 dispatches per handle, in order): scratch/r5_landing_pmc.sh."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CZ_TABLE_SETTLE"] = "0"
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch


def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k, B, ef = int(os.environ.get("HS_N", 10_000_000)), 768, 10, 1024, 144
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=64)
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)

    def timed(ix, tag):
        run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(4): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(24): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 24
        print(f"HANDLE {tag} ms {ms:.3f} frac {int(nd.sum().item()) * 4 * dim / ms / 1e6 / 8000:.3f}", flush=True)

    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    timed(ix, "built")
    nodes, nbrs, entry = ix.export()
    ix.close()
    torch.cuda.empty_cache()
    MB = 1 << 20
    for i, js in enumerate([0, 3 * MB + 4096, 0, 5 * 1024 * MB + 4096, 0, 37 * 1024 * MB + 12288, 0, 0]):
        junk = torch.empty(js, dtype=torch.uint8, device=dev) if js else None
        ix2 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
        del junk
        torch.cuda.empty_cache()
        timed(ix2, f"created{i}")
        ix2.close()
        torch.cuda.empty_cache()


main()

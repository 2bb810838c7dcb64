"""scratch (round 6): latency of small batches with and without the speculative step (CZ_HNSW_SPEC), one index.
HS_N (default 1M), HS_EF (96 at 1M, 144 at 10M), HS_BS; results compared bit for bit (ids, distances, counts, n_dist)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k = int(os.environ.get("HS_N", 1_000_000)), 768, 10
    ef = int(os.environ.get("HS_EF", 96 if n <= 1_000_000 else 144))
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=int(os.environ.get("HS_EFC", 200)))
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"n={n} ef={ef}: build {time.time() - t0:.1f}s", flush=True)
    del x
    torch.cuda.empty_cache()
    qall = Bn.gen_vectors(torch, 1024, dim, "lowrank", 43, dev)
    for B in [int(b) for b in os.environ.get("HS_BS", "1,2,8,32,64,128,256").split(",")]:
        q = qall[:B].contiguous()
        res = {}
        for spec in ("0", "1"):
            os.environ["CZ_HNSW_SPEC"] = spec
            ids = torch.empty((B, k), dtype=torch.int32, device=dev)
            dd = torch.empty((B, k), dtype=torch.float64, device=dev)
            cnt = torch.empty(B, dtype=torch.int32, device=dev)
            nd = torch.zeros(B, dtype=torch.int64, device=dev)
            run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)  # noqa: E731
            for _ in range(3): run()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            res[spec] = (best, ids, dd, cnt, nd)
        same = all(bool(torch.equal(a, b)) for a, b in zip(res["0"][1:], res["1"][1:]))
        nd = float(res["1"][4].to(torch.float64).mean().item())
        print(f"B={B:4d}: plain {res['0'][0]:.3f} ms   speculative {res['1'][0]:.3f} ms  ({res['0'][0] / res['1'][0]:.2f} x)   same results: {same}   n_dist/q {nd:.0f}", flush=True)
    os.environ.pop("CZ_HNSW_SPEC", None)
    ix.close()
main()

"""scratch (round 6): the pending buffer in front of a large list (search_level_pending) against the plain step (CZ_HNSW_PEND=0) on a
clustered HS_N x 768 index: ms per 1 024-query batch at ef 2048 / 4096 / 8192, results must be bit-identical."""
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
    n, dim, k, B = int(os.environ.get("HS_N", 1_000_000)), 768, 10, int(os.environ.get("HS_B", 1024))
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, "clustered", 42, dev)
    q = Bn.gen_vectors(torch, B, dim, "clustered", 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"build {time.time() - t0:.1f}s", flush=True)
    del x
    torch.cuda.empty_cache()
    ids = torch.empty((B, k), dtype=torch.int32, device=dev)
    dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    nd = torch.zeros(B, dtype=torch.int64, device=dev)
    for ef in (1024, 2048, 4096, 8192):
        ref = None
        for pend in ("0", "1"):
            os.environ["CZ_HNSW_PEND"] = pend
            def run():
                ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tot = int(nd.sum().item())
            cur = (ids.clone(), dd.clone(), nd.clone())
            same = None if ref is None else bool(torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1]) and torch.equal(ref[2], cur[2]))
            if ref is None: ref = cur
            print(f"ef={ef:5d} pend={pend}: {ms:8.2f} ms  {B / ms * 1e3:8.0f} q/s  {tot * 4 * dim / ms / 1e6 / 8000:.3f} of peak  n_dist/q {tot / B:.0f}  same_as_plain={same}", flush=True)
    ix.close()
main()

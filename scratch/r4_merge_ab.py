"""scratch (round 4): merge_wide before / after (one process per library: COZO_GPU_LIB), one clustered index, ef ladder; prints the time
per batch and a digest of (ids, distances, evaluation counts) so that two runs can be compared line by line."""
import hashlib, os, sys, time
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
    n, dim, k, B = int(os.environ.get("HS_N", 1_000_000)), 768, 10, 1024
    dist = os.environ.get("HS_DIST", "clustered")
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, dist, 42, dev)
    q = Bn.gen_vectors(torch, B, dim, dist, 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
    for ef in [int(e) for e in os.environ.get("HS_EF", "144,600,768,1024,2048,4096,8192").split(",")]:
        kk = min(k, ef)
        ids = torch.empty((B, kk), dtype=torch.int32, device=dev)
        dd = torch.empty((B, kk), dtype=torch.float64, device=dev)
        cnt = torch.empty(B, dtype=torch.int32, device=dev)
        nd = torch.zeros(B, dtype=torch.int64, device=dev)
        def run():
            ix.hnsw_knn_batch_device(q, HnswSearch(k=kk, ef=ef), ids, dd, cnt, nd, stream)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5 if ef <= 1024 else 3
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot = int(nd.sum().item())
        h = hashlib.sha1(ids.cpu().numpy().tobytes() + dd.cpu().numpy().tobytes() + nd.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{dist} n={n} ef={ef:5d}: {ms:8.3f} ms  {B / ms * 1e3:8.0f} q/s  {tot * 4 * dim / ms / 1e6 / 8000:.3f} of peak  n_dist/q {tot / B:.0f}  digest {h}", flush=True)
    ix.close()
main()

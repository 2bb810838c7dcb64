"""scratch (round 5): hnsw_knn_kernel against the batch size per launch -- latency / throughput ladder on one 1M x 768 index (ef 96),
persistent grid + rows-in-flight chosen from B (product build) or the round-4 library (COZO_GPU_LIB=scratch/lib/libcozo_gpu_r4.so).
Results of the wide variants (U = 4 / 8) are compared with the U = 2 kernel's bit for bit."""
import os, sys, time, json
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
    ef = int(os.environ.get("HS_EF", 96))
    kind = os.environ.get("HS_DIST", "lowrank")
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, kind, 42, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"lib {os.environ.get('COZO_GPU_LIB', 'product')}: build {time.time() - t0:.1f}s", flush=True)
    del x
    torch.cuda.empty_cache()
    BMAX = 8192
    qall = Bn.gen_vectors(torch, BMAX, dim, kind, 43, dev)
    check = os.environ.get("HS_CHECK", "1") == "1"
    for B in [int(b) for b in os.environ.get("HS_BS", "1,8,64,256,512,768,1024,1280,2048,4096,8192").split(",")]:
        q = qall[:B].contiguous()
        ids = torch.empty((B, k), dtype=torch.int32, device=dev)
        dd = torch.empty((B, k), dtype=torch.float64, device=dev)
        cnt = torch.empty(B, dtype=torch.int32, device=dev)
        nd = torch.zeros(B, dtype=torch.int64, device=dev)
        def run():
            ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(3): run()
        torch.cuda.synchronize()
        best = 1e9
        reps = 20 if B <= 1024 else 6
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        tot = float(nd.to(torch.float64).sum().item())
        same = ""
        if check:
            a = (ids.clone(), dd.clone(), cnt.clone(), nd.clone())
            os.environ["CZ_HNSW_U"] = "2"
            run(); torch.cuda.synchronize()
            os.environ.pop("CZ_HNSW_U")
            same = "  == U=2 kernel: " + str(all(bool(torch.equal(u, v)) for u, v in zip(a, (ids, dd, cnt, nd))))
        print(f"B={B:5d} ef={ef}: {best:.3f} ms  {B / best * 1e3:8.0f} q/s  {tot * 4 * dim / best / 1e6 / 8000:.3f} of peak   n_dist/q {tot / B:.0f}{same}", flush=True)
    ix.close()
main()

"""scratch: per-phase time of hnsw_knn_kernel's level-0 loop (profiling build, scratch/build_prof.sh)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["COZO_GPU_LIB"] = os.path.join(ROOT, "scratch", "lib", "libcozo_gpu_prof.so")
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k, B = int(os.environ.get("PH_N", 1_000_000)), 768, 10, int(os.environ.get("PH_B", 1024))
    kind = os.environ.get("PH_DIST", "lowrank")
    efs = [int(v) for v in os.environ.get("PH_EF", "96").split(",")]
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, kind, 42, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
    q = Bn.gen_vectors(torch, B, dim, kind, 43, dev)
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)
    for ef in efs:
        run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(2): run()
        buf = (C.c_ulonglong * 12)()
        L.cz_debug_phase_cycles(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        L.cz_debug_phase_cycles(buf, 1)
        v = [buf[i] / reps / B for i in range(12)]  # per query
        names = ["select+mark", "row+visited", "eval rounds", "finish", "merge (rest)", None, None, None, "merge: compact", "merge: rank", "merge: shift"]
        tot = sum(v[:5]) + sum(v[8:11])
        steps, rows = v[5], v[6]
        print(f"ef={ef}: kernel {ms:.3f} ms/batch; per query: {steps:.1f} steps, {rows:.0f} rows evaluated ({rows/steps:.1f}/step); "
              f"timed cycles/query {tot:.0f}; {tot / steps:.0f} cycles per step")
        for i, nme in enumerate(names):
            if nme is None:
                continue
            print(f"  {nme:12s} {v[i]:12.0f} cycles/query  {100*v[i]/tot:5.1f} %   {v[i]/steps:8.0f} cycles/step")
main()

"""scratch (round 5): cz_knn_bruteforce(CZ_BF_GEMM) with the selection fused into the GEMM's epilogue against the unfused form
(CZ_BF_FUSE=0): time per 1024-query batch over N x 768 vectors, results equal"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np, torch, ctypes as C
import bench as Bn
from cozo_amd._lib import ptr, check
dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
n, dim, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 768, 1024, 10
x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
# a vector-only index: one level, no links
xh = x.cpu().numpy()
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
ix = GpuHnswIndex(HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=8), xh, [None], [np.full((n, 1), 0xFFFFFFFF, dtype=np.uint32)], 0)
del x, xh
ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
res = {}
for fuse in ("0", "1", "0", "1"):
    os.environ["CZ_BF_FUSE"] = fuse
    check(L.cz_knn_bruteforce(ix._h, ptr(q), B, k, ptr(ids), ptr(dd), _lib.CZ_DEVICE_PTRS | _lib.CZ_BF_GEMM, None))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(L.cz_knn_bruteforce(ix._h, ptr(q), B, k, ptr(ids), ptr(dd), _lib.CZ_DEVICE_PTRS | _lib.CZ_BF_GEMM, None))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[fuse] = (ids.clone(), dd.clone())
    print(f"n={n} fuse={fuse}: {dt * 1e3:.1f} ms per {B} queries = {B / dt:.0f} q/s, {2.0 * B * n * dim / dt / 1e12:.1f} TFLOP/s", flush=True)
print("fused == unfused:", bool(torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])))

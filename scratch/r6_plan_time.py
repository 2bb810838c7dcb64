"""host build time of the in-place plan on the bench-sized uniform graph (CPU only)"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from tests.test_inplace_plan import build
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rng = np.random.default_rng(4242)
dst = rng.integers(0, N, E, dtype=np.int64); src = rng.integers(0, N, E, dtype=np.int64)
k = np.unique(dst[src != dst] * N + src[src != dst]); del dst, src
d = k // N; s = (k - d * N).astype(np.uint32); del k
ioff = np.zeros(N + 1, np.uint64); ioff[1:] = np.cumsum(np.bincount(d, minlength=N)); del d
od = np.bincount(s, minlength=N).astype(np.uint32)
L = C.CDLL(build())
L.ipt_emulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
sc = np.empty(N, np.float32); info = np.zeros(8, np.uint64); err = C.c_double(0)
for sweeps in (0, 1):
    t0 = time.time()
    rc = L.ipt_emulate(ioff.ctypes.data, s.ctypes.data, od.ctypes.data, N, 4096, 256, 16384, 16384, 1, 0.85, sweeps, 0, sc.ctypes.data, C.byref(err), info.ctypes.data)
    print("rc", rc, "sweeps", sweeps, f"{time.time() - t0:.1f}s", info)

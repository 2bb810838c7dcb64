"""scratch (round 2): the vertex-partitioned BFS / SSSP entry points with ONE rank (collectives skipped) on the 10M / 100M graph beside
the one-GPU rules: what the partitioned formulation itself costs on a device, and that its rows are the one-GPU rule's.
Run in a child process that leaves through os._exit (two users of the RCCL library in one process, tests/gpu_comm_child.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
from cozo_amd import graph as G
from cozo_amd import comm as CM
dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
n, e = 10_000_000, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
keep = src != dst
key = torch.unique(src[keep] * n + dst[keep])
s = torch.div(key, n, rounding_mode="floor"); t = key - s * n
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
ooff, otgt = off.to(torch.int32).cpu().numpy().view(np.uint32), t.to(torch.int32).cpu().numpy().view(np.uint32)
w = (torch.randint(1, 64, (otgt.size,), generator=g, device=dev, dtype=torch.int32).to(torch.float32) / 8).cpu().numpy()
del src, dst, keep, key, s, t, off
torch.cuda.empty_cache()
comm = CM.Comm(CM.Comm.unique_id(), 0, 1)
starts = np.array([0], dtype=np.uint32)
def timed(name, fn):
    fn(); t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print(f"{name:40s} {dt * 1e3:8.1f} ms wall", flush=True)
    return r
p1, d1, _, _ = timed("cz_bfs", lambda: G.bfs(ooff, otgt, starts, want_depth=True))
p2, d2, _, _ = timed("cz_bfs_sharded (1 rank)", lambda: CM.bfs_sharded(comm, ooff, otgt, n, 0, n, starts, want_depth=True))
print("  bfs rows equal:", bool(np.array_equal(p1, p2) and np.array_equal(d1, d2)))
a1, b1 = timed("cz_sssp", lambda: G.sssp(ooff, otgt, w, starts))
a2, b2 = timed("cz_sssp_sharded (1 rank)", lambda: CM.sssp_sharded(comm, ooff, otgt, w, n, 0, n, starts))
print("  sssp costs equal:", bool(np.array_equal(a1, a2)), " parents equal:", bool(np.array_equal(b1, b2)), flush=True)
os._exit(0)

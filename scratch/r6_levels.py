"""Level structure of the ascending Gauss-Seidel sweep on the bench's uniform graph (CPU, numpy): how many rows and edges per level,
and how many of a level's in-edges come from the level just before it (the ones on the critical path)."""
import sys, time
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rng = np.random.default_rng(4242)
dst = rng.integers(0, N, E, dtype=np.int64)
src = rng.integers(0, N, E, dtype=np.int64)
k = np.unique(dst[src != dst] * N + src[src != dst])
dst = (k // N).astype(np.int32); src = (k - (k // N) * N).astype(np.int32); del k
E = len(dst)
fw = src < dst
fs, fd = src[fw], dst[fw]
print("edges", E, "forward", len(fs))
# rows of forward edges are grouped by dst (sorted): reduceat over row starts
starts = np.flatnonzero(np.r_[True, fd[1:] != fd[:-1]])
rows = fd[starts]
level = np.zeros(N, dtype=np.int32)
t0 = time.time()
for r in range(200):
    cand = np.maximum.reduceat(level[fs] + 1, starts)
    if np.array_equal(level[rows], cand):
        break
    level[rows] = cand
print("levels", level.max() + 1, "rounds", r, f"{time.time()-t0:.0f}s")
L = level.max() + 1
cnt = np.bincount(level, minlength=L)
indeg = np.bincount(dst, minlength=N)
edges_in = np.bincount(level, weights=indeg, minlength=L)
gap = level[fd] - level[fs]
urgent1 = np.bincount(level[fd][gap == 1], minlength=L)
urgent2 = np.bincount(level[fd][gap <= 2], minlength=L)
urgent4 = np.bincount(level[fd][gap <= 4], minlength=L)
fwd_in = np.bincount(level[fd], minlength=L)
print("lvl rows in_edges fwd_in gap1 gap<=2 gap<=4")
for l in range(L):
    print(l, cnt[l], int(edges_in[l]), fwd_in[l], urgent1[l], urgent2[l], urgent4[l])
print("total gap1", urgent1.sum(), "gap<=2", urgent2.sum(), "gap<=4", urgent4.sum(), "of fwd", len(fs))
np.save("/tmp/r6_level.npy", level)

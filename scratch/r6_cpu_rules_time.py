"""how long the oracle takes per traversal rule at the bench's size (CPU only; numpy graph of the same statistics)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rng = np.random.default_rng(7)
src = rng.integers(0, n, e, dtype=np.int64); dst = rng.integers(0, n, e, dtype=np.int64)
key = np.unique(src[src != dst] * n + dst[src != dst]); del src, dst
s = key // n; t = (key - s * n).astype(np.uint32)
off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum(np.bincount(s, minlength=n))
E = t.size
w = (rng.integers(1, 64, E) / 8).astype(np.float32)
key2 = np.sort(np.concatenate([key, t.astype(np.int64) * n + s])); del key
s2 = key2 // n; t2 = (key2 - s2 * n).astype(np.uint32)
off2 = np.zeros(n + 1, np.uint64); off2[1:] = np.cumsum(np.bincount(s2, minlength=n)); del key2, s2, s
print("graph ready", E, t2.size, flush=True)
for name, fn in (("bfs", lambda: O.bfs_order(n, off, t, 0)), ("cc", lambda: O.tarjan_groups(n, off2, t2)),
                 ("sssp", lambda: O.dijkstra(n, off, t, w, 0)), ("triangles", lambda: O.clustering_coefficients(n, off2, t2)),
                 ("lp_colouring", lambda: O.lp_colouring(n, off2, t2)),
                 ("lp", lambda: O.label_propagation(n, off2, t2, np.ones(t2.size, np.float32), 10))):
    t0 = time.time(); fn(); print(name, f"{time.time() - t0:.1f}s", flush=True)

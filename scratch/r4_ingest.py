"""scratch (round 4): host ingest at scale with stage times (CZI_TRACE=1): rows, nodes from argv"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cozo_amd import build as B, codec
from cozo_amd.ingest import StoredGraph
B.build_ingest()
n_rows, n_nodes = int(sys.argv[1]), int(sys.argv[2])
cache = f"/tmp/ingest_rec_{n_rows}_{n_nodes}.npy"
if os.path.exists(cache):
    rec = np.load(cache)
else:
    rng = np.random.default_rng(9)
    key = np.unique(rng.integers(0, n_nodes, n_rows, dtype=np.int64) * n_nodes + rng.integers(0, n_nodes, n_rows, dtype=np.int64))
    # the relation is sorted by KEY BYTES; node ids are scattered over the value range so that first appearances are not ascending
    perm = rng.permutation(n_nodes).astype(np.int64)
    a, b = perm[key // n_nodes], perm[key % n_nodes]
    order = np.lexsort((b, a))
    a, b = a[order], b[order]
    e = a.size
    rec = np.zeros((e, 28), dtype=np.uint8)
    rec[:, 7] = 1
    for c, col in enumerate((a, b)):
        img = col.astype(np.float64).view(np.uint64) | np.uint64(0x8000000000000000)
        rec[:, 8 + 10 * c] = 0x05
        rec[:, 9 + 10 * c:17 + 10 * c] = img.byteswap().view(np.uint8).reshape(e, 8)
    np.save(cache, rec)
e = rec.shape[0]
rows = codec.StoredRows(rec.tobytes(), np.arange(e + 1, dtype=np.uint64) * 28, b"", np.zeros(e + 1, dtype=np.uint64), 2)
for rep in range(2):
    t0 = time.perf_counter()
    g = StoredGraph(rows)
    t1 = time.perf_counter()
    o1 = g.csr(False)
    o2 = g.csr(True)
    t2 = time.perf_counter()
    print(f"rows {e} nodes {g.n}: ids {t1 - t0:.3f}s csr both {t2 - t1:.3f}s -> {e / (t2 - t0) / 1e6:.2f} M rows/s", flush=True)
    import hashlib
    print("digest", hashlib.sha1(o1[0].tobytes() + o1[1].tobytes() + o2[0].tobytes() + o2[1].tobytes()).hexdigest()[:16], flush=True)
    g.close()

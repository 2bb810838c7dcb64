"""scratch: throughput of czi_hnsw_ingest on the stored bytes of a synthetic `tbl:idx` + base relation (one core).
   python scratch/ingest_hnsw_bench.py [n_nodes] [dim] [m]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cozo_amd import build as B, codec  # noqa: E402
from cozo_amd.ingest import StoredHnswIndex, encode_index_rows  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
B.build_ingest()
rng = np.random.default_rng(0)
vecs = rng.random((n, dim), dtype=np.float32)
# a random regular "index": level 0 with 2m links per node, one upper level with every 16th node and m links
nb0 = np.sort(rng.integers(0, n, (n, 2 * m)).astype(np.uint32), axis=1)
up = np.arange(0, n, 16, dtype=np.uint32)
nb1 = np.sort(up[rng.integers(0, up.size, (up.size, m))], axis=1)
for tab, ids in ((nb0, np.arange(n)), (nb1, up)):
    for r in range(tab.shape[0]):  # no self links, no duplicates
        row = np.unique(tab[r][tab[r] != ids[r]])
        tab[r] = 0xFFFFFFFF
        tab[r, :row.size] = row
key_of_node = [(int(i), 1, -1) for i in range(n)]
dist = [rng.random(nb0.shape), rng.random(nb1.shape)]
t_enc = 1e9
for _ in range(2):
    t0 = time.time()
    idx = encode_index_rows(key_of_node, vecs, [None, up], [nb0, nb1], 0, 0, dist, 2)
    t_enc = min(t_enc, time.time() - t0)
base = codec.StoredRows.from_tuples(1, [(int(i), vecs[i]) for i in range(n)], 1)
best = 1e9
for _ in range(3):
    t0 = time.time()
    got = StoredHnswIndex(idx, base, [1], dim, 0, m)
    best = min(best, time.time() - t0)
assert got.n == n and np.array_equal(got.vectors, vecs) and np.array_equal(got.level_nbrs[0][:, :nb0.shape[1]], nb0)
mb = (len(idx.keys) + len(idx.vals) + len(base.keys) + len(base.vals)) / 1e6
print(f"{n} nodes, dim {dim}, m {m}: {len(idx)} index rows + {len(base)} base rows = {mb:.0f} MB of stored bytes "
      f"; write-back czi_hnsw_encode_rows (incl. the Python-side key list) {t_enc:.3f} s = {len(idx) / t_enc / 1e6:.1f} M rows/s; "
      f"ingest incl. copying the arrays out to numpy {best:.3f} s = {len(idx) / best / 1e6:.1f} M index rows/s, {mb / best:.0f} MB/s")

import sys, numpy as np
sys.path.insert(0,'.')
from tests import util
from oracle import oracle as O
from cozo_amd import _lib
from cozo_amd.hnsw import HnswSearch
_lib.lib().cz_init(0)
y = util.vectors(200, 16, 5, "normal")
y[10] = y[3]
y[77] = 0.0
_, flat = util.build_index(O, y, 1, 6, 30)
gc = util.gpu_index(flat, "Cosine", 6)
ids, dist, cnt = gc.hnsw_knn_batch(y[:32], HnswSearch(k=8, ef=40))
oids, odist, ocnt, _ = flat.knn_batch(y[:32], 8, 40, dot_mode=O.DOT_GPU)
for b in range(32):
    if not (np.array_equal(ids[b], oids[b]) and np.array_equal(np.nan_to_num(dist[b],nan=-1), np.nan_to_num(odist[b],nan=-1))):
        print(b, cnt[b], ocnt[b]); print(ids[b]); print(oids[b]); print(dist[b]); print(odist[b])

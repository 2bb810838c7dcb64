import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cozo_amd import _lib
L = _lib.lib()
assert L.cz_init(0) == 0
from oracle import oracle
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
dim, dist, metric = int(sys.argv[1]), "L2", 0
B = int(sys.argv[2])
n, m = 3000, 8
rng = np.random.default_rng(dim)
x = rng.standard_normal((n, dim))
b = oracle.HnswBuilder(dim, metric, m, 40)
b.insert(x.astype(np.float32), oracle.random_levels(n, m, 3))
f32flat = b.export()
print("built", flush=True)
flat = oracle.FlatIndex(x, metric, f32flat.level_nodes, f32flat.level_nbrs, f32flat.entry, f64=True)
q = rng.standard_normal((B, dim))
oids, odd, ocnt, ond = flat.knn_batch(q, 10, 32, dot_mode=oracle.DOT_GPU)
print("oracle knn ok", oids[0][:3], odd[0][:3], flush=True)
man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, dtype="F64")
g = GpuHnswIndex(man, x, [None] + flat.level_nodes[1:], flat.level_nbrs, flat.entry)
print("gpu index ok", flush=True)
ids, dd, cnt, nd = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=32), with_n_dist=True)
print("gpu knn ok", ids[0][:3], dd[0][:3], np.array_equal(ids, oids), np.array_equal(dd, odd), int(nd.sum()), ond, flush=True)

"""scratch (round 5): recall of batched builds against the sequential build (max_batch = 1, the reference's hnsw_put order) on one corpus"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cozo_amd import _lib
L = _lib.lib()
assert L.cz_init(0) == 0
from tests import util
from oracle import oracle as O
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
n, dim, m, efc = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, 64, 16, 100
x = util.vectors(n, dim, 31, "lowrank")
q = util.vectors(512, dim, 32, "lowrank")
levels = O.random_levels(n, m, 8)
man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=m, ef_construction=efc)
gt = None
for mb in [int(v) for v in os.environ.get("BQ_BATCHES", "1,256,1024,4096").split(",")]:
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, levels=levels, max_batch=mb)
    dt = time.time() - t0
    if gt is None:
        gt, _ = ix.bruteforce_knn(q, 10)
    rec = []
    for ef in (16, 32, 64, 128):
        ids, _, _ = ix.hnsw_knn_batch(q, HnswSearch(k=10, ef=ef))
        rec.append(float(np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(q))])))
    print(f"n={n} max_batch={mb:5d}: build {dt:6.1f}s  recall@10 at ef 16/32/64/128: " + " ".join(f"{r:.4f}" for r in rec), flush=True)
    ix.close()

"""batched extended build: does it run, is it deterministic"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import util
from oracle import oracle
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest

n, mb = int(sys.argv[1]), int(sys.argv[2])
ext = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
dim, m, efc = 48, 8, 40
x = util.vectors(n, dim, 23, "lowrank")
levels = oracle.random_levels(n, m, 6)
man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=efc, extend_candidates=ext)
g = GpuHnswIndex.build(man, x, levels=levels, max_batch=mb)
a = g.export()[1]
diffs = []
for rep in range(3):
    g2 = GpuHnswIndex.build(man, x, levels=levels, max_batch=mb)
    b = g2.export()[1]
    diffs.append([int((u != v).any(axis=1).sum()) for u, v in zip(a, b)])
    g2.close()
print(f"n={n} max_batch={mb} extend={ext}: n_dist {g.last_build_n_dist}; rows differing from the first build, per level: {diffs}", flush=True)

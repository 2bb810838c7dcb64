"""first divergence of the extended device build from the oracle (max_batch = 1)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import util
from oracle import oracle
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest

n, dim, dist, metric, m, efc, keep, kind = 500, 24, "L2", 0, 4, 20, False, "uniform"
if len(sys.argv) > 1:
    keep = bool(int(sys.argv[1]))
x = util.vectors(n, dim, 13, kind)
levels = oracle.random_levels(n, m, 4)
man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, ef_construction=efc, extend_candidates=True,
                        keep_pruned_connections=keep)
for k in list(range(2, 80)) + [100, 150, 200, 300, 500]:
    b = oracle.HnswBuilder(dim, metric, m, efc, extend_candidates=True, keep_pruned_connections=keep, dot_mode=oracle.DOT_GPU)
    b.insert(x[:k], levels[:k])
    flat = b.export()
    g = GpuHnswIndex.build(man, x[:k], levels=levels[:k], max_batch=1)
    nodes, nbrs, entry = g.export()
    deg = g.degrees()
    bad = False
    for lv in range(flat.n_levels):
        want = flat.level_nbrs[lv]
        wdeg = np.array([b.degree(int(v), lv) for v in flat.level_nodes[lv]])
        if not np.array_equal(nbrs[lv], want) or not np.array_equal(deg[lv], wdeg):
            bad = True
            rows = np.nonzero((nbrs[lv] != want).any(axis=1) | (deg[lv] != wdeg))[0]
            print(f"k={k} level {lv}: {len(rows)} rows differ (levels of the last vector: {levels[k-1]})")
            for r in rows[:6]:
                print("  node", int(nodes[lv][r]), "gpu", [int(v) for v in nbrs[lv][r] if v != 0xFFFFFFFF], "deg", deg[lv][r],
                      "| oracle", [int(v) for v in want[r] if v != 0xFFFFFFFF], "deg", wdeg[r])
    g.close()
    if bad:
        break
else:
    print("no divergence")

"""GPU parity: index construction (hnsw_put_vector / select_neighbours_heuristic / shrink) vs the oracle.

max_batch = 1 is the reference's sequential algorithm: the link tables must be IDENTICAL to the oracle's
(oracle in the kernels' summation order).  Larger batches differ only in that vectors of one batch do not see
each other; there the bar is structural invariants + search quality on the same queries."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SEQ_CASES = [
    (700, 24, "L2", 0, 6, 30, False, "uniform"),
    (500, 100, "Cosine", 1, 8, 40, False, "lowrank"),
    (400, 768, "Cosine", 1, 4, 24, True, "lowrank"),
    (600, 130, "IP", 2, 5, 25, False, "normal"),
]


@pytest.mark.parametrize("n,dim,dist,metric,m,efc,keep,kind", SEQ_CASES)
def test_sequential_build_identical_to_oracle(gpu_lib, oracle, n, dim, dist, metric, m, efc, keep, kind):
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    x = util.vectors(n, dim, 11, kind)
    levels = oracle.random_levels(n, m, 3)
    b = oracle.HnswBuilder(dim, metric, m, efc, keep_pruned_connections=keep, dot_mode=oracle.DOT_GPU)
    b.insert(x, levels)
    flat = b.export()
    man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, ef_construction=efc,
                            keep_pruned_connections=keep)
    g = GpuHnswIndex.build(man, x, levels=levels, max_batch=1)
    nodes, nbrs, entry = g.export()
    assert entry == flat.entry and len(nbrs) == flat.n_levels
    for lv in range(flat.n_levels):
        assert np.array_equal(nodes[lv], flat.level_nodes[lv])
        assert np.array_equal(nbrs[lv], flat.level_nbrs[lv]), f"level {lv} link rows differ"
    assert g.last_build_n_dist > 0
    assert np.array_equal(g.export_vectors(), x)


def test_batched_build_quality_and_invariants(gpu_lib, oracle):
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, dim, m, efc = 20000, 64, 12, 80
    x = util.vectors(n, dim, 21, "lowrank")
    q = util.vectors(200, dim, 22, "lowrank")
    levels = oracle.random_levels(n, m, 5)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=efc)
    g = GpuHnswIndex.build(man, x, levels=levels, max_batch=1024)
    nodes, nbrs, entry = g.export()
    # structure: level populations from `levels`, widths m_max0 / m_max, ascending rows, no self links, ids in range
    for lv in range(len(nbrs)):
        want = np.nonzero(levels >= lv)[0].astype(np.uint32)
        assert np.array_equal(nodes[lv], want)
        tab = nbrs[lv].astype(np.int64)
        assert tab.shape[1] == (2 * m if lv == 0 else m)
        live = tab != 0xFFFFFFFF
        assert (tab[live] < n).all() and not (tab == nodes[lv][:, None]).any()
        tab[~live] = 2 ** 40
        assert (np.diff(tab, axis=1) > 0).all() or (np.diff(tab, axis=1)[np.diff(tab, axis=1) <= 0] == 0).all()
        if lv > 0:  # links stay inside the level
            assert np.isin(nbrs[lv][nbrs[lv] != 0xFFFFFFFF], nodes[lv]).all()
    assert entry == int(np.argmax(levels)) and levels[entry] == len(nbrs) - 1
    assert (nbrs[0] != 0xFFFFFFFF).sum(axis=1).min() >= 1
    # the oracle searches the GPU-built graph and the GPU searches it: identical (same index, same arithmetic)
    flat = oracle.FlatIndex(x, oracle.L2, nodes, nbrs, entry)
    ids, dist, cnt = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=64))
    oids, odist, ocnt, _ = flat.knn_batch(q, 10, 64, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids, oids) and np.array_equal(dist, odist)
    # quality: recall within a whisker of the sequentially built (oracle) index at the same ef
    gt, _ = g.bruteforce_knn(q, 10)
    rec_gpu = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(q))])
    b = oracle.HnswBuilder(dim, oracle.L2, m, efc)
    b.insert(x, levels)
    sids, _, _, _ = b.export().knn_batch(q, 10, 64)
    rec_seq = np.mean([len(set(sids[i]) & set(gt[i])) / 10 for i in range(len(q))])
    assert rec_gpu >= rec_seq - 0.03, (rec_gpu, rec_seq)


def test_build_edge_cases(gpu_lib, oracle):
    from cozo_amd import _lib
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    man = HnswIndexManifest(vec_dim=8, distance="L2", m_neighbours=4, ef_construction=10)
    e = GpuHnswIndex.build(man, np.zeros((0, 8), np.float32))
    ids, _, cnt = e.hnsw_knn_batch(np.ones((2, 8), np.float32), HnswSearch(k=3, ef=5))
    assert (cnt == 0).all()
    one = GpuHnswIndex.build(man, np.ones((1, 8), np.float32))
    ids, dist, cnt = one.hnsw_knn_batch(np.ones((1, 8), np.float32), HnswSearch(k=3, ef=5))
    assert cnt[0] == 1 and ids[0, 0] == 0 and dist[0, 0] == 0
    # seeded levels drawn inside the library are reproducible
    x = util.vectors(300, 8, 2)
    a = GpuHnswIndex.build(man, x, seed=9, max_batch=1).export()
    b = GpuHnswIndex.build(man, x, seed=9, max_batch=1).export()
    assert all(np.array_equal(u, v) for u, v in zip(a[1], b[1]))
    with pytest.raises(_lib.CozoGpuError):
        GpuHnswIndex.build(HnswIndexManifest(vec_dim=8, m_neighbours=4, extend_candidates=True), x)
    with pytest.raises(_lib.CozoGpuError):
        GpuHnswIndex.build(HnswIndexManifest(vec_dim=8, m_neighbours=200), x)

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
    (1500, 20, "L2", 0, 6, 1200, False, "uniform"),  # ef_construction beyond round 3's 1 024: the ranked merge in the build kernels
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
        GpuHnswIndex.build(HnswIndexManifest(vec_dim=8, m_neighbours=200), x)


@pytest.mark.parametrize("n,dim,dist,metric,m,efc,keep,kind", SEQ_CASES[:3])
def test_insert_into_existing_index_identical_to_one_sequential_build(gpu_lib, oracle, n, dim, dist, metric, m, efc, keep, kind):
    """cz_hnsw_insert = hnsw_put on a later write (stored.rs:431-450 -> hnsw.rs:679-727).  With max_batch = 1 the tables
    after build(first part) + insert(second part) are the tables of ONE sequential build over all rows -- both when the first
    part was built on the device and when it was uploaded from the store (its link distances are then evaluated again on
    the way back into build form, and must be the bits the original insertion stored)."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    x = util.vectors(n, dim, 11, kind)
    levels = oracle.random_levels(n, m, 3)
    h = n * 2 // 3
    b = oracle.HnswBuilder(dim, metric, m, efc, keep_pruned_connections=keep, dot_mode=oracle.DOT_GPU)
    b.insert(x[:h], levels[:h])
    part = b.export()
    b.insert(x[h:], levels[h:])
    flat = b.export()
    man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, ef_construction=efc, keep_pruned_connections=keep)
    built = GpuHnswIndex.build(man, x[:h], levels=levels[:h], max_batch=1)
    uploaded = GpuHnswIndex(man, part.vectors, [None] + part.level_nodes[1:], part.level_nbrs, part.entry)
    for g in (built, uploaded):
        g.insert(x[h:], levels=levels[h:], max_batch=1)
        nodes, nbrs, entry = g.export()
        assert entry == flat.entry and len(nbrs) == flat.n_levels
        for lv in range(flat.n_levels):
            assert np.array_equal(nodes[lv], flat.level_nodes[lv])
            assert np.array_equal(nbrs[lv], flat.level_nbrs[lv]), f"level {lv} link rows differ"
        assert np.array_equal(g.export_vectors(), x)
        g.close()


def test_remove_then_insert(gpu_lib, oracle):
    """cz_hnsw_remove (hnsw_remove, hnsw.rs:728-868): the nodes leave every level, no link names them any more, the entry
    point moves to the smallest node of the highest level left; the search over what remains equals the oracle's search
    over the same tables.  A later insert (rows renumbered underneath) keeps all of that."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, dim, m, efc = 4000, 48, 8, 40
    x = util.vectors(n + 500, dim, 31, "lowrank")
    q = util.vectors(64, dim, 32, "lowrank")
    levels = oracle.random_levels(n + 500, m, 7)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=m, ef_construction=efc)
    g = GpuHnswIndex.build(man, x[:n], levels=levels[:n], max_batch=256)
    nodes0, nbrs0, entry0 = g.export()
    rng = np.random.default_rng(5)
    dead = np.unique(np.concatenate([rng.choice(n, 300, replace=False), [entry0], np.nonzero(levels[:n] >= 2)[0]])).astype(np.uint32)
    g.remove(dead)

    def check(g, n_now, lv_now, dead):
        nodes, nbrs, entry = g.export()
        alive_lv = np.where(np.isin(np.arange(n_now), dead), -1, lv_now[:n_now])
        assert len(nbrs) == alive_lv.max() + 1
        assert entry == int(np.nonzero(alive_lv == alive_lv.max())[0][0])
        for lv in range(len(nbrs)):
            tab = nbrs[lv]
            live = tab != 0xFFFFFFFF
            assert not np.isin(tab[live], dead).any(), "a link still names a removed node"
            if lv == 0:
                assert not live[dead].any(), "a removed node still holds links"
            else:
                assert np.array_equal(nodes[lv], np.nonzero(alive_lv >= lv)[0].astype(np.uint32))
            # live links first, ascending
            t = tab.astype(np.int64)
            t[~live] = 2 ** 40
            assert (np.diff(t, axis=1) >= 0).all()
        return nodes, nbrs, entry

    nodes, nbrs, entry = check(g, n, levels, dead)
    # what survived of every row is exactly the old row minus the removed ids
    for lv in range(len(nbrs)):
        keep_rows = np.isin(nodes0[lv], nodes[lv]) if lv else np.ones(n, bool)
        old = nbrs0[lv][keep_rows]
        for r in range(0, old.shape[0], 97):
            want = [v for v in old[r].tolist() if v != 0xFFFFFFFF and v not in set(dead.tolist())]
            if lv == 0 and r in set(dead.tolist()):
                want = []
            assert nbrs[lv][r][:len(want)].tolist() == want
    flat = oracle.FlatIndex(x[:n], oracle.COSINE, nodes, nbrs, entry)
    ids, dist, cnt = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=50))
    oids, odist, ocnt, _ = flat.knn_batch(q, 10, 50, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids, oids) and np.array_equal(dist, odist) and np.array_equal(cnt, ocnt)
    assert not np.isin(ids[ids != 0xFFFFFFFF], dead).any()
    # insert after remove: the upper-level rows are renumbered, the removed nodes stay gone
    g.insert(x[n:], levels=levels[n:], max_batch=64)
    nodes, nbrs, entry = check(g, n + 500, levels, dead)
    flat = oracle.FlatIndex(x, oracle.COSINE, nodes, nbrs, entry)
    ids, dist, cnt = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=50))
    oids, odist, ocnt, _ = flat.knn_batch(q, 10, 50, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids, oids) and np.array_equal(dist, odist)
    gt, _ = g.bruteforce_knn(q, 10)  # (the exhaustive scan still sees the removed rows' vectors: compare on the living)
    rec = np.mean([len(set(ids[i]) & (set(gt[i]) - set(dead.tolist()))) / max(1, len(set(gt[i]) - set(dead.tolist()))) for i in range(len(q))])
    assert rec >= 0.8
    g.close()


def test_write_back_of_insert_and_remove_is_a_delta(oracle, gpu_lib):
    """Index maintenance on the device, written back as the rows that changed: the `tbl:idx` rows of the index
    after cz_hnsw_insert / cz_hnsw_remove against the rows the store holds -- codec.stored_rows_delta -- are a small part of the
    relation, and applying them to the stored rows gives exactly the rows of the new index; read back (libcozo_ingest) it
    searches like the device index it came from."""
    from cozo_amd import build as B, codec
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    from cozo_amd.ingest import StoredHnswIndex
    B.build_ingest()
    rng = np.random.default_rng(12)
    n0, n1, dim, m = 3000, 40, 24, 8
    vecs = rng.standard_normal((n0 + n1, dim)).astype(np.float32)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=40)
    levels = oracle.random_levels(n0 + n1, m, 5)
    ix = GpuHnswIndex.build(man, vecs[:n0], levels=levels[:n0], max_batch=1)
    keys = [(f"k{i:05d}", 1, -1) for i in range(n0 + n1)]
    stored = ix.index_rows(keys[:n0], relation_id=12)
    # ---- insert: the new nodes' rows + the rows of the nodes whose lists they entered (and whatever those shrinks dropped)
    ix.insert(vecs[n0:], levels=levels[n0:], max_batch=1)
    after = ix.index_rows(keys, relation_id=12)
    puts, dels = codec.stored_rows_delta(stored, after)
    assert 0 < len(puts) < len(after) // 10 and len(dels) < len(after) // 50
    stored = codec.apply_stored_delta(stored, puts, dels)
    assert stored.keys == after.keys and stored.vals == after.vals
    # ---- remove: the rows of the removed nodes go, the rows that pointed at them go
    gone = np.array([7, 500, n0 + 3], dtype=np.uint32)
    ix.remove(gone)
    after = ix.index_rows(keys, relation_id=12)
    puts, dels = codec.stored_rows_delta(stored, after)
    assert len(dels) > 0 and len(puts) + len(dels) < len(after) // 10
    stored = codec.apply_stored_delta(stored, puts, dels)
    assert stored.keys == after.keys and stored.vals == after.vals
    base = codec.StoredRows.from_tuples(11, [(k[0], vecs[i]) for i, k in enumerate(keys) if i not in set(gone.tolist())], 1)
    back = StoredHnswIndex(stored, base, [1], dim, oracle.L2, m).to_gpu(man)
    q = rng.standard_normal((32, dim)).astype(np.float32)
    a = ix.hnsw_knn_batch(q, HnswSearch(k=10, ef=40))
    b = back.hnsw_knn_batch(q, HnswSearch(k=10, ef=40))
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])  # the same distances and counts (ids are renumbered by the scan)
    ix.close()
    back.close()


@pytest.mark.gpu
def test_remove_equals_the_restated_hnsw_remove(gpu_lib, oracle):
    """cz_hnsw_remove against the oracle's restatement of hnsw_remove_vec (hnsw.rs:754-868) on the SAME index: after the same
    removals the device holds exactly the reference's rows minus the ones the reference leaves dangling (rows of other nodes
    that still name a removed node, which the reference can no longer follow: ensure_key fails on them, :133).  That is the
    one deliberate difference, and the count of such rows is reported.  Node lists per level, link rows, entry point: equal."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, dim, m, efc = 3000, 32, 8, 40
    x = util.vectors(n, dim, 41, "lowrank")
    levels = oracle.random_levels(n, m, 11)
    for dist, metric in (("L2", oracle.L2), ("Cosine", oracle.COSINE)):
        b = oracle.HnswBuilder(dim, metric, m, efc, dot_mode=oracle.DOT_GPU)
        b.insert(x, levels)
        flat = b.export()
        man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, ef_construction=efc)
        g = GpuHnswIndex(man, flat.vectors, [None] + flat.level_nodes[1:], flat.level_nbrs, flat.entry)
        rng = np.random.default_rng(6)
        dead = np.unique(np.concatenate([rng.choice(n, 400, replace=False), [flat.entry], flat.level_nodes[-1]])).astype(np.uint32)
        g.remove(dead)
        assert b.remove(dead.tolist()) == len(dead)
        want = b.export()
        nodes, nbrs, entry = g.export()
        assert entry == want.entry and len(nbrs) == want.n_levels
        alive0 = want.level_nodes[0]
        for lv in range(want.n_levels):
            if lv == 0:  # the device keeps a (now empty) level-0 row per removed id: ids are not renumbered
                assert (nbrs[0][dead] == 0xFFFFFFFF).all()
                got_rows = nbrs[0][alive0]
            else:
                assert np.array_equal(nodes[lv], want.level_nodes[lv])
                got_rows = nbrs[lv]
            w = want.level_nbrs[lv].shape[1]
            assert (got_rows[:, w:] == 0xFFFFFFFF).all()
            assert np.array_equal(got_rows[:, :w], want.level_nbrs[lv]), f"{dist}: level {lv} rows differ from hnsw_remove's"
        print(f"{dist}: rows the reference leaves dangling after {len(dead)} removals: {b.dangling_links()}")
        q = util.vectors(32, dim, 42, "lowrank")
        ids, dd, cnt = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=50))
        oids, odd, ocnt, _ = want.knn_batch(q, 10, 50, dot_mode=oracle.DOT_GPU)
        # (ids of the oracle's export are positions in ITS node list at level 0 == node ids, since ids are kept)
        assert np.array_equal(ids, oids) and np.array_equal(dd, odd) and np.array_equal(cnt, ocnt)
        g.close()


@pytest.mark.gpu
def test_entry_point_is_the_smallest_key_on_the_top_layer(gpu_lib, oracle):
    """The reference's entry point is positional: the first row of the index relation = the smallest KEY on the top layer
    (hnsw.rs:184-191, 891-899).  Rows inserted later get node ids n, n+1, ... whatever their keys are, so insert / remove
    compare key ranks (cz_hnsw_set_key_order), not ids: a later row on the top layer whose key sorts first becomes the entry
    point, for the inserts after it and for every search -- as in the oracle's restatement, table for table."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n0, n1, dim, m, efc = 1500, 400, 24, 8, 40
    x = util.vectors(n0 + n1, dim, 51, "lowrank")
    levels = oracle.random_levels(n0 + n1, m, 13)
    top0 = int(levels[:n0].max())
    levels[n0:] = np.minimum(levels[n0:], top0)
    levels[n0 + 5] = top0   # two later rows on the top layer: one with the smallest key of all, one with a large key
    levels[n0 + 9] = top0
    rng = np.random.default_rng(14)
    keys = np.concatenate([1000 + 2 * np.arange(n0), rng.permutation(np.arange(1001, 1001 + 2 * n1, 2))])
    keys[n0 + 5] = 3        # sorts before everything
    keys[n0 + 9] = 10 ** 6  # sorts behind everything
    assert len(set(keys.tolist())) == n0 + n1
    rank = np.argsort(np.argsort(keys)).astype(np.uint32)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=efc)
    b = oracle.HnswBuilder(dim, oracle.L2, m, efc, dot_mode=oracle.DOT_GPU)
    b.insert(x[:n0], levels[:n0])
    entry_before = b.export().entry
    b.set_key_order(rank)
    b.insert(x[n0:], levels[n0:])
    want = b.export()
    assert want.entry == n0 + 5 != entry_before
    g = GpuHnswIndex.build(man, x[:n0], levels=levels[:n0], max_batch=1)
    g.insert(x[n0:], levels=levels[n0:], max_batch=1, key_rank=rank)
    nodes, nbrs, entry = g.export()
    assert entry == want.entry
    for lv in range(want.n_levels):
        assert np.array_equal(nodes[lv], want.level_nodes[lv])
        assert np.array_equal(nbrs[lv][:, :want.level_nbrs[lv].shape[1]], want.level_nbrs[lv]), f"level {lv}"
    q = util.vectors(16, dim, 52, "lowrank")
    ids, dd, cnt = g.hnsw_knn_batch(q, HnswSearch(k=5, ef=30))
    oids, odd, ocnt, _ = want.knn_batch(q, 5, 30, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids, oids) and np.array_equal(dd, odd)
    # removing the entry point: the next smallest key on the top layer takes over, on both sides
    g.remove([entry])
    b.remove([entry])
    top_nodes = [int(v) for v in np.nonzero(levels == top0)[0] if v != entry]
    expect = min(top_nodes, key=lambda v: rank[v])
    assert g.export()[2] == b.export().entry == expect
    g.close()


EXT_CASES = [
    (500, 24, "L2", 0, 4, 20, False, "uniform"),
    (360, 100, "Cosine", 1, 6, 16, True, "lowrank"),
    (300, 768, "Cosine", 1, 4, 12, False, "lowrank"),
    (400, 33, "IP", 2, 3, 10, True, "normal"),
]


def _oracle_degrees(b, flat):
    return [np.array([b.degree(int(v), lv) for v in flat.level_nodes[lv]]) for lv in range(flat.n_levels)]


@pytest.mark.parametrize("n,dim,dist,metric,m,efc,keep,kind", EXT_CASES)
def test_sequential_build_with_extend_candidates_identical_to_oracle(gpu_lib, oracle, n, dim, dist, metric, m, efc, keep, kind):
    """extend_candidates (hnsw.rs:499-511) with max_batch = 1: the neighbours' neighbours join the heuristic's candidates
    (a scratch array in global memory, sorted, chunked through the LDS list), one neighbour after the other gets its reverse
    link and its shrink, and a shrink that selects the target itself leaves a degree one above the row's links (:413-433,
    :352-357; pinned for the oracle by tests/literal_hnsw_store.py).  Tables AND degrees equal the oracle's; so do the tables
    of build(first part) + insert(second part), which carries the self-link flags across the two calls."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    x = util.vectors(n, dim, 13, kind)
    levels = oracle.random_levels(n, m, 4)
    b = oracle.HnswBuilder(dim, metric, m, efc, extend_candidates=True, keep_pruned_connections=keep, dot_mode=oracle.DOT_GPU)
    b.insert(x, levels)
    flat = b.export()
    want_deg = _oracle_degrees(b, flat)
    man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, ef_construction=efc, extend_candidates=True,
                            keep_pruned_connections=keep)
    whole = GpuHnswIndex.build(man, x, levels=levels, max_batch=1)
    h = n * 2 // 3
    parts = GpuHnswIndex.build(man, x[:h], levels=levels[:h], max_batch=1)
    parts.insert(x[h:], levels=levels[h:], max_batch=1)
    self_links = 0
    for g in (whole, parts):
        nodes, nbrs, entry = g.export()
        assert entry == flat.entry and len(nbrs) == flat.n_levels
        deg = g.degrees()
        for lv in range(flat.n_levels):
            assert np.array_equal(nodes[lv], flat.level_nodes[lv])
            assert np.array_equal(nbrs[lv], flat.level_nbrs[lv]), f"level {lv} link rows differ"
            assert np.array_equal(deg[lv], want_deg[lv]), f"level {lv} degrees differ"
            self_links += int((deg[lv] - (nbrs[lv] != 0xFFFFFFFF).sum(axis=1)).sum())
        g.close()
    assert self_links > 0  # the quirk was exercised
    # and it is a different index from the one built without the extension
    plain = oracle.HnswBuilder(dim, metric, m, efc, keep_pruned_connections=keep, dot_mode=oracle.DOT_GPU)
    plain.insert(x, levels)
    assert not np.array_equal(plain.export().level_nbrs[0], flat.level_nbrs[0])


def test_batched_build_with_extend_candidates(gpu_lib, oracle, monkeypatch):
    """Batched: shrinks of one round read the rows as the round found them (staged selections); a reverse link that waited
    for room and was meanwhile picked up by the row's shrink is not entered twice.  Structure and search quality as for the
    plain batched build; with eager shrinking (no row ever waits for room) two builds give the same tables."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, dim, m, efc = 12000, 48, 8, 40
    x = util.vectors(n, dim, 23, "lowrank")
    q = util.vectors(200, dim, 24, "lowrank")
    levels = oracle.random_levels(n, m, 6)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=efc, extend_candidates=True)
    monkeypatch.setenv("CZ_BUILD_LAZY", "0")
    # (32 vectors per batch: a row gains at most one link per inserted vector, i.e. never more than its 32 slack slots in a
    # round, so no reverse link ever waits for room and nothing depends on the order the atomics were served in)
    e1 = GpuHnswIndex.build(man, x[:4000], levels=levels[:4000], max_batch=32)
    e2 = GpuHnswIndex.build(man, x[:4000], levels=levels[:4000], max_batch=32)
    assert all(np.array_equal(a, c) for a, c in zip(e1.export()[1], e2.export()[1]))
    assert all(np.array_equal(a, c) for a, c in zip(e1.degrees(), e2.degrees()))
    monkeypatch.delenv("CZ_BUILD_LAZY")
    g = GpuHnswIndex.build(man, x, levels=levels, max_batch=512)
    nodes, nbrs, entry = g.export()
    for lv in range(len(nbrs)):
        tab = nbrs[lv]
        assert tab.shape[1] == (2 * m if lv == 0 else m)
        live = tab != 0xFFFFFFFF
        assert (tab[live] < n).all() and not (tab == nodes[lv][:, None]).any()
        srt = np.sort(tab, axis=1)
        assert not ((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 0xFFFFFFFF)).any()  # no link twice
        extra = g.degrees()[lv] - live.sum(axis=1)
        assert ((extra == 0) | (extra == 1)).all()
        if lv > 0:
            assert np.isin(tab[live], nodes[lv]).all()
    ids, dist, cnt = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=64))
    flat = oracle.FlatIndex(x, oracle.L2, nodes, nbrs, entry)
    oids, odist, _, _ = flat.knn_batch(q, 10, 64, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids, oids) and np.array_equal(dist, odist)
    gt, _ = g.bruteforce_knn(q, 10)
    rec = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(q))])
    plain = GpuHnswIndex.build(HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=efc), x, levels=levels,
                               max_batch=512)
    pids, _, _ = plain.hnsw_knn_batch(q, HnswSearch(k=10, ef=64))
    rec_plain = np.mean([len(set(pids[i]) & set(gt[i])) / 10 for i in range(len(q))])
    assert rec >= rec_plain - 0.05, (rec, rec_plain)


def test_extend_candidates_staging_that_does_not_fit_is_a_clear_oom(gpu_lib, oracle, monkeypatch):
    """a round of shrinks is staged whole; when the staging does not fit the device the build says so (CZ_E_OOM, with the
    number of rows that would fit) instead of a bare allocation failure.  The cap is forced through CZ_BUILD_STAGE_CAP_BYTES."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    n, dim, m = 6000, 32, 8
    x = util.vectors(n, dim, 29, "lowrank")
    levels = oracle.random_levels(n, m, 7)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=32, extend_candidates=True)
    monkeypatch.setenv("CZ_BUILD_STAGE_CAP_BYTES", "4096")  # a handful of rows
    with pytest.raises(Exception) as ei:
        GpuHnswIndex.build(man, x, levels=levels, max_batch=512)
    assert "staging" in str(ei.value) and "rows per" in str(ei.value)
    monkeypatch.delenv("CZ_BUILD_STAGE_CAP_BYTES")
    g = GpuHnswIndex.build(man, x, levels=levels, max_batch=512)  # the library is fine afterwards
    assert g.export()[1][0].shape[0] == n


def test_write_back_with_extend_candidates_carries_the_degrees(oracle, gpu_lib):
    """The self rows written back hold the reference's degree (cz_hnsw_index_export_degrees -> czi_hnsw_encode_rows_degrees): with
    extend_candidates one above the node's link rows wherever a shrink selected the node itself."""
    from cozo_amd import build as B
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    B.build_ingest()
    n, dim, m = 400, 16, 4
    vecs = util.vectors(n, dim, 3)
    levels = oracle.random_levels(n, m, 8)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=16, extend_candidates=True)
    ix = GpuHnswIndex.build(man, vecs, levels=levels, max_batch=1)
    b = oracle.HnswBuilder(dim, oracle.L2, m, 16, extend_candidates=True, dot_mode=oracle.DOT_GPU)
    b.insert(vecs, levels)
    keys = [(i, 1, -1) for i in range(n)]
    tup = ix.index_rows(keys, relation_id=3).tuples()
    selfs = {(-t[0], t[1]): t[7] for t in tup if t[0] <= 0 and t[1:4] == t[4:7]}
    links = {}
    for t in tup:
        if t[0] <= 0 and t[1:4] != t[4:7]:
            links[(-t[0], t[1])] = links.get((-t[0], t[1]), 0) + 1
    assert len(selfs) == int((levels + 1).sum())
    above = 0
    for (lv, node), deg in selfs.items():
        assert deg == b.degree(node, lv)
        above += int(deg) - links.get((lv, node), 0)
    assert above > 0
    ix.close()


@pytest.mark.parametrize("extend,keep,dist,metric", [(False, False, "L2", 0), (False, True, "Cosine", 1), (True, False, "L2", 0),
                                                     (True, True, "IP", 2)])
def test_rows_with_several_vectors_identical_to_oracle(gpu_lib, oracle, extend, keep, dist, metric):
    """Rows that carry several indexed vectors (hnsw.rs:694-706): links inside one base row are written and counted into the
    degrees, and hnsw_get_neighbours never returns them (:609-610).  cz_hnsw_set_row_of + max_batch = 1: the visible tables and
    the degrees equal the oracle's (pinned by tests/literal_hnsw_store.py), for one build and for build + insert, and no row
    holds a link into its own base row."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    n, dim, m, efc = 420, 20, 4, 16
    rng = np.random.default_rng(7)
    row_of = np.sort(rng.integers(0, n // 3, n)).astype(np.uint32)
    base = rng.standard_normal((n // 3, dim)).astype(np.float32)
    x = (base[row_of] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)  # a row's vectors are each other's nearest
    levels = oracle.random_levels(n, m, 5)
    b = oracle.HnswBuilder(dim, metric, m, efc, extend_candidates=extend, keep_pruned_connections=keep, dot_mode=oracle.DOT_GPU)
    b.set_row_of(row_of)
    b.insert(x, levels)
    flat = b.export()
    want_deg = _oracle_degrees(b, flat)
    man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, ef_construction=efc, extend_candidates=extend,
                            keep_pruned_connections=keep)
    whole = GpuHnswIndex.build(man, x, levels=levels, max_batch=1, row_of=row_of)
    h = n * 2 // 3
    parts = GpuHnswIndex.build(man, x[:h], levels=levels[:h], max_batch=1, row_of=row_of[:h])
    parts.insert(x[h:], levels=levels[h:], max_batch=1, row_of=row_of)
    hidden = 0
    for g in (whole, parts):
        nodes, nbrs, entry = g.export()
        assert entry == flat.entry and len(nbrs) == flat.n_levels
        deg = g.degrees()
        for lv in range(flat.n_levels):
            assert np.array_equal(nodes[lv], flat.level_nodes[lv])
            assert np.array_equal(nbrs[lv], flat.level_nbrs[lv]), f"level {lv} link rows differ"
            assert np.array_equal(deg[lv], want_deg[lv]), f"level {lv} degrees differ"
            live = nbrs[lv] != 0xFFFFFFFF
            fr = np.broadcast_to(nodes[lv][:, None], nbrs[lv].shape)
            assert not (row_of[nbrs[lv][live]] == row_of[fr[live]]).any()
            hidden += int((deg[lv] - live.sum(axis=1)).sum())
        g.close()
    assert hidden > 0
    # batched: the same rule, structurally -- also with ONE row of 250 vectors (its members select each other all the time: the
    # links that are counted and not kept pile up on them until a shrink forgets them)
    big = row_of.copy()
    big[100:350] = big[100]
    for rows in (row_of, big):
        g = GpuHnswIndex.build(man, x, levels=levels, max_batch=64, row_of=rows)
        nodes, nbrs, _ = g.export()
        deg = g.degrees()
        for lv in range(len(nbrs)):
            live = nbrs[lv] != 0xFFFFFFFF
            fr = np.broadcast_to(nodes[lv][:, None], nbrs[lv].shape)
            assert not (rows[nbrs[lv][live]] == rows[fr[live]]).any()
            extra = deg[lv] - live.sum(axis=1)
            assert (extra >= 0).all() and extra.max() < 64
        g.close()


def test_batched_build_reaches_the_sequential_builds_recall(gpu_lib, oracle):
    """the bench's index is a max_batch = 4096 build, a graph the reference never produces (only max_batch = 1 is
    the reference's sequential hnsw_put, pinned to the oracle).  What a batched build must keep is the QUALITY of that graph.
    Measured (profiles/r05_build_quality.txt: 30k / 60k / 200k vectors, batches of 256 ... 4096): where the sequential build's
    recall@10 is >= 0.99 the batched build is within 0.2 points at the same ef, around 0.95-0.98 within a point, at ef 16 two to
    three points lower -- half of it the lazy shrinking of a batched round, the rest the batch itself.  The bounds here are those
    measurements with some room, on a corpus small enough for the sequential build to take seconds."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, dim, m, efc = 20000, 64, 16, 100
    x = util.vectors(n, dim, 31, "lowrank")
    q = util.vectors(512, dim, 32, "lowrank")
    levels = oracle.random_levels(n, m, 8)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=m, ef_construction=efc)
    seq = GpuHnswIndex.build(man, x, levels=levels, max_batch=1)
    bat = GpuHnswIndex.build(man, x, levels=levels, max_batch=4096)
    gt, _ = seq.bruteforce_knn(q, 10)

    def recall(ix, ef):
        ids, _, _ = ix.hnsw_knn_batch(q, HnswSearch(k=10, ef=ef))
        return float(np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(q))]))
    for ef, room in ((16, 0.04), (32, 0.015), (64, 0.005)):
        rs, rb = recall(seq, ef), recall(bat, ef)
        assert rb >= rs - room, (ef, rs, rb)
    assert recall(bat, 64) >= 0.99


def test_extend_candidates_round_larger_than_the_old_staging(gpu_lib, oracle, monkeypatch):
    """ADVICE r3 / VERDICT r4 #6: a round of more shrinks than the staging arrays held (65 536 until round 5) was applied in
    parts, a later part reading rows an earlier one had rewritten.  The staging now grows to the round (it starts empty, so every
    build with extend_candidates exercises the growth; the final lazy round of this one holds thousands of rows) and the
    structure invariants hold."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    n, dim, m, efc = 6000, 32, 8, 40
    x = util.vectors(n, dim, 41, "lowrank")
    levels = oracle.random_levels(n, m, 9)
    man = HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=efc, extend_candidates=True)
    a = GpuHnswIndex.build(man, x, levels=levels, max_batch=2048)
    nodes, nbrs, entry = a.export()
    for lv in range(len(nbrs)):
        tab = nbrs[lv]
        live = tab != 0xFFFFFFFF
        assert (tab[live] < n).all() and not (tab == nodes[lv][:, None]).any()
        srt = np.sort(tab, axis=1)
        assert not ((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 0xFFFFFFFF)).any()

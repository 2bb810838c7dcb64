"""CPU tests of the oracle itself: reference-pinned golden cases, hand-computable known answers and
independent cross-checks (numpy f64 / scipy).  No GPU needed."""
import json
import os

import numpy as np
import pytest

from tests import util

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_distance_known_answers(oracle):
    O = oracle
    # mirrors runtime/tests.rs:691-697: l2_dist([1,2],[2,3]) = 2 ; cos_dist(v,v) = 0 ; ip_dist of unit vector = 0
    assert O.distance(O.L2, [1, 2], [2, 3]) == 2.0
    assert O.distance(O.COSINE, [1, 2], [1, 2]) == pytest.approx(0.0, abs=1e-7)
    assert O.distance(O.IP, [0.6, 0.8], [0.6, 0.8]) == pytest.approx(0.0, abs=1e-7)
    assert np.isnan(O.distance(O.COSINE, [0, 0], [1, 2]))


@pytest.mark.parametrize("dim", [1, 2, 7, 8, 9, 15, 16, 17, 128, 768, 1536])
def test_distance_vs_float64(oracle, dim):
    O = oracle
    rng = np.random.default_rng(dim)
    a = rng.standard_normal(dim).astype(np.float32)
    b = rng.standard_normal(dim).astype(np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    exact = {O.L2: np.sum((a64 - b64) ** 2), O.IP: 1 - a64 @ b64,
             O.COSINE: 1 - a64 @ b64 / np.sqrt((a64 @ a64) * (b64 @ b64))}
    for metric, ref in exact.items():
        for mode in (O.DOT_NDARRAY, O.DOT_GPU):
            got = O.distance(metric, a, b, mode)
            assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref), float(np.sum(np.abs(a64 * b64))))


def test_unrolled_dot_order(oracle):
    """ndarray's unrolled_dot: 8 lanes, (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7), then the tail -- checked against a
    literal numpy float32 transcription."""
    rng = np.random.default_rng(3)
    for n in [0, 1, 7, 8, 9, 23, 64, 100, 771]:
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        p = np.zeros(8, np.float32)
        i = 0
        while n - i >= 8:
            p = (p + a[i:i + 8] * b[i:i + 8]).astype(np.float32)
            i += 8
        s = np.float32(0)
        for j in range(4):
            s = np.float32(s + np.float32(p[j] + p[j + 4]))
        for j in range(i, n):
            s = np.float32(s + np.float32(a[j] * b[j]))
        assert oracle.dot_ndarray(a, b) == float(s)


def test_love_graph_golden(oracle):
    """the `love` graph of algos/shortest_path_bfs.rs:124-174: alice->bob has length 3, alice->george is Null."""
    O = oracle
    g = json.load(open(os.path.join(GOLD, "love_graph.json")))
    names = sorted({x for e in g["edges"] for x in e})
    key = {nm: i for i, nm in enumerate(names)}  # memcmp order of strings = lexicographic
    rows = sorted((key[a], key[b]) for a, b in g["edges"])
    fi, ti, ind = O.assign_ids([r[0] for r in rows], [r[1] for r in rows])
    inv = {int(k): i for i, k in enumerate(ind)}
    off, tgt = O.build_csr(len(ind), fi, ti)
    for case in g["expect"]:
        s, t = inv[key[case["from"]]], inv.get(key[case["to"]], None)
        parent = O.shortest_path_bfs(len(ind), off, tgt, s, [t] if t is not None else [])
        path = O.path_from_parent(parent, s, t) if t is not None else None
        if case["len"] is None:
            assert path is None
        else:
            assert len(path) == case["len"]
            assert [names[int(ind[p])] for p in path][0] == case["from"]


def test_csr_layout_sorted_with_duplicates(oracle):
    O = oracle
    src = np.array([2, 0, 0, 1, 0], np.uint32)
    dst = np.array([1, 2, 1, 0, 1], np.uint32)  # duplicate 0->1 kept (CsrLayout::Sorted, not Deduplicated)
    off, tgt = O.build_csr(3, src, dst)
    assert off.tolist() == [0, 3, 4, 5] and tgt.tolist() == [1, 1, 2, 0, 1]
    off, tgt = O.build_csr(3, src, dst, undirected=True)
    assert off.tolist() == [0, 4, 8, 10] and tgt.tolist() == [1, 1, 1, 2, 0, 0, 0, 2, 0, 1]


def test_first_appearance_ids(oracle):
    fi, ti, ind = oracle.assign_ids([10, 10, 20, 5], [20, 5, 5, 10])
    assert ind.tolist() == [10, 20, 5] and fi.tolist() == [0, 0, 1, 2] and ti.tolist() == [1, 2, 2, 0]


def test_pagerank_vs_float64_power_iteration(oracle):
    import scipy.sparse as sp
    O = oracle
    frm, to = util.random_relation(3000, 20000, 11)
    g = util.graph_from_relation(O, frm, to)
    n = g["n"]
    s, it, err = O.pagerank(n, g["ioff"], g["isrc"], g["outdeg"], 0.85, 0.0, 15)
    s8, it8, _ = O.pagerank(n, g["ioff"], g["isrc"], g["outdeg"], 0.85, 0.0, 15, threads=4)
    assert it == it8 == 15 and np.array_equal(s, s8)
    a = sp.csr_matrix((np.ones(len(g["fi"])), (g["ti"], g["fi"])), shape=(n, n))
    x = np.full(n, 1.0 / n)
    od = g["outdeg"].astype(np.float64)
    for _ in range(15):
        c = np.where(od > 0, x / np.where(od > 0, od, 1), 0)
        x = (1 - 0.85) / n + 0.85 * (a @ c)
    assert np.max(np.abs(s - x) / x) < 5e-6
    # no dangling-mass redistribution: with sinks the scores do not sum to one
    assert (g["outdeg"] == 0).any() and s.sum() < 1.0


def test_tarjan_groups_vs_scipy(oracle):
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cg
    O = oracle
    frm, to = util.random_relation(4000, 3500, 4)
    g = util.graph_from_relation(O, frm, to, undirected=True)
    n = g["n"]
    grp, k = O.tarjan_groups(n, g["ooff"], g["otgt"])
    ncomp, lab = cg.connected_components(sp.csr_matrix((np.ones(len(g["fi"])), (g["fi"], g["ti"])), shape=(n, n)),
                                         directed=False)
    first = np.full(ncomp, n)
    np.minimum.at(first, lab, np.arange(n))
    assert k == ncomp and np.array_equal(grp, np.argsort(np.argsort(first))[lab])
    gd = util.graph_from_relation(O, frm, to)
    _, kd = O.tarjan_groups(gd["n"], gd["ooff"], gd["otgt"])
    nscc, _ = cg.connected_components(sp.csr_matrix((np.ones(len(gd["fi"])), (gd["fi"], gd["ti"])),
                                                    shape=(gd["n"], gd["n"])), directed=True, connection="strong")
    assert kd == nscc
    # a 300k-node chain must not overflow (the reference recurses; the oracle uses an explicit stack)
    c = np.arange(0, 299999, dtype=np.int64)
    gc = util.graph_from_relation(O, c, c + 1, undirected=True)
    assert O.tarjan_groups(gc["n"], gc["ooff"], gc["otgt"])[1] == 1


def test_bfs_and_dijkstra_vs_scipy(oracle):
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cg
    O = oracle
    frm, to = util.random_relation(1500, 6000, 8)
    w = np.random.default_rng(8).random(len(frm)).astype(np.float32)
    g = util.graph_from_relation(O, frm, to, weights=w)
    n = g["n"]
    rows = np.repeat(np.arange(n), np.diff(g["ooff"]).astype(int))
    m = sp.csr_matrix((g["ow"].astype(np.float64), (rows, g["otgt"])), shape=(n, n))
    hops = cg.shortest_path(m, unweighted=True, indices=0)
    parent = O.shortest_path_bfs(n, g["ooff"], g["otgt"], 0, np.arange(n))
    for t in range(1, n):
        p = O.path_from_parent(parent, 0, t)
        assert (p is None) == np.isinf(hops[t]) and (p is None or len(p) - 1 == hops[t])
    assert O.path_from_parent(parent, 0, 0) is None  # start == end yields Null in the reference
    dist, par = O.dijkstra(n, g["ooff"], g["otgt"], g["ow"], 0)
    dd = cg.dijkstra(m, indices=0)
    fin = np.isfinite(dd)
    assert np.array_equal(np.isfinite(dist), fin)
    assert np.max(np.abs(dist[fin] - dd[fin]) / np.maximum(dd[fin], 1e-9)) < 1e-5


def test_hnsw_build_invariants_and_recall(oracle):
    O = oracle
    x = util.vectors(3000, 32, 1, "lowrank")
    b, flat = util.build_index(O, x, O.L2, 8, 60)
    assert flat.n_levels >= 2 and flat.level_size[0] == 3000
    # level populations shrink geometrically; every upper-level node exists below
    for lv in range(1, flat.n_levels):
        assert flat.level_size[lv] < flat.level_size[lv - 1]
        assert np.isin(flat.level_nodes[lv], flat.level_nodes[lv - 1]).all()
    assert flat.entry in flat.level_nodes[-1]
    # live degrees respect m_max0 / m_max, rows ascending, no self links
    for lv in range(flat.n_levels):
        tab = flat.level_nbrs[lv].astype(np.int64)
        assert tab.shape[1] == (16 if lv == 0 else 8)
        tab[tab == O.NONE] = 2 ** 40
        assert (np.diff(tab, axis=1) >= 0).all()
        assert not (tab == flat.level_nodes[lv][:, None]).any()
    q = util.vectors(100, 32, 2, "lowrank")
    gt, _ = O.bruteforce_knn(O.L2, x, q, 10)
    ids, dist, cnt, nd = flat.knn_batch(q, 10, 100)
    rec = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(100)])
    assert rec > 0.9 and (cnt == 10).all() and (np.diff(dist, axis=1) >= 0).all()
    # both summation orders walk the same graph: near-identical results
    ids2, dist2, _, _ = flat.knn_batch(q, 10, 100, dot_mode=O.DOT_GPU)
    assert (ids == ids2).mean() > 0.98


def test_hnsw_radius_empty_and_filter_width(oracle):
    O = oracle
    x = util.vectors(500, 16, 3)
    _, flat = util.build_index(O, x, O.L2, 6, 30)
    ids, dist, cnt, _ = flat.knn_batch(x[:5], 5, 20, radius=0.0)
    assert (cnt == 1).all() and (ids[:, 0] == np.arange(5)).all() and (dist[:, 0] == 0).all()
    empty = O.FlatIndex(np.zeros((0, 16), np.float32), O.L2, [], [], O.NONE)
    ids, dist, cnt, _ = empty.knn_batch(x[:3], 5, 20)
    assert (cnt == 0).all()


def test_clustering_coefficients_against_matrix_powers(oracle):
    """triangles.rs:70-110 on SIMPLE undirected graphs: triangles(v) = (A^3)_vv / 2, degree = row sum; plus a
    hand-checked multigraph (a doubled edge doubles the pairs it takes part in)."""
    rng = np.random.default_rng(7)
    for n, p in [(12, 0.5), (60, 0.15), (150, 0.05)]:
        A = np.triu((rng.random((n, n)) < p).astype(np.int64), 1)
        A = A + A.T
        f, t = np.nonzero(A)
        off, tgt = oracle.build_csr(n, f, t)
        cc, tri, deg = oracle.clustering_coefficients(n, off, tgt)
        want_tri = np.diag(np.linalg.matrix_power(A, 3)) // 2
        want_deg = A.sum(1)
        assert np.array_equal(tri.astype(np.int64), want_tri) and np.array_equal(deg.astype(np.int64), want_deg)
        want_cc = np.where(want_deg >= 2, 2.0 * want_tri / np.maximum(want_deg * (want_deg - 1.0), 1.0), 0.0)
        assert np.array_equal(cc, want_cc)
    # triangle 0-1-2 with the edge 0-1 present twice: node 2 sees one pair (1 > 0), nodes 0 and 1 see their
    # doubled neighbour twice
    f = np.array([0, 1, 0, 1, 1, 2, 0, 2]); t = np.array([1, 0, 1, 0, 2, 1, 2, 0])
    off, tgt = oracle.build_csr(3, f, t)
    cc, tri, deg = oracle.clustering_coefficients(3, off, tgt)
    assert deg.tolist() == [3, 3, 2] and tri.tolist() == [2, 2, 1]


def test_label_propagation_fixed_order_is_a_fixpoint_of_the_reference_rule(oracle):
    """orc_label_propagation_in_order (label_propagation.rs:56-109 with the order and the tie-break handed in): when the loop
    stops before max_iter, every node's label is one of the best-scored labels among its out-neighbours (what :77-91 leaves
    behind), whatever order was handed in; and the colouring is proper (no edge inside a class, either direction)."""
    import numpy as np
    rng = np.random.default_rng(3)
    n = 300
    frm = rng.integers(0, n, 1500)
    to = (frm // 30) * 30 + rng.integers(0, 30, frm.size)  # ten planted groups
    frm, to = np.concatenate([frm, to]), np.concatenate([to, frm])
    order = np.lexsort((to, frm))
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(np.bincount(frm, minlength=n))
    tgt = to[order].astype(np.uint32)
    w = (rng.integers(1, 5, tgt.size) / 2).astype(np.float32)
    colour, k = oracle.lp_colouring(n, off, tgt)
    src = np.repeat(np.arange(n), np.diff(off).astype(np.int64))
    assert k >= 2 and (colour[src[src != tgt]] != colour[tgt[src != tgt]]).all()
    for node_order in (np.arange(n, dtype=np.uint32), rng.permutation(n).astype(np.uint32),
                       np.lexsort((np.arange(n), colour)).astype(np.uint32)):
        labels, it = oracle.label_propagation_in_order(n, off, tgt, w, node_order, 50)
        assert it < 50
        for v in range(n):
            lo, hi = int(off[v]), int(off[v + 1])
            if lo == hi:
                assert labels[v] == v
                continue
            score = {}
            for e in range(lo, hi):
                lab = int(labels[tgt[e]])
                score[lab] = np.float32(score.get(lab, np.float32(0.0)) + w[e])
            best = max(score.values())
            assert labels[v] == min(lab for lab, sc in score.items() if sc == best)
    assert len(np.unique(labels)) <= 40


def test_hnsw_remove_restated(oracle):
    """orc_hnsw_remove = hnsw_remove_vec (hnsw.rs:754-868): the node's rows go at every layer, so does the reverse row of every
    node it linked to (live or soft-deleted), that neighbour's stored degree drops by one EVEN IF it held no live link back;
    rows of other nodes that still name the removed one are left behind (counted; a reference search following one fails)."""
    rng = np.random.default_rng(3)
    x = rng.random((1500, 12), dtype=np.float32)
    levels = oracle.random_levels(1500, 6, 4)
    b = oracle.HnswBuilder(12, oracle.L2, 6, 30)
    b.insert(x, levels)
    before = b.export()
    links_before = {(l, int(i), int(t)) for l in range(before.n_levels) for r, i in enumerate(before.level_nodes[l])
                    for t in before.level_nbrs[l][r] if t != oracle.NONE}
    deg_before = {(l, int(i)): b.degree(i, l) for l in range(before.n_levels) for i in before.level_nodes[l]}
    dead = sorted({int(before.entry), 5, 17, 300, 301, 1499} | {int(i) for i in before.level_nodes[-1]})
    assert b.remove(dead) == len(dead)
    assert b.remove([5]) == 0  # no self row any more: nothing to do (:766-778 breaks at layer 0)
    after = b.export()
    alive_top = np.where(np.isin(np.arange(1500), dead), -1, levels)
    assert after.n_levels == alive_top.max() + 1
    assert after.entry == int(np.nonzero(alive_top == alive_top.max())[0][0])  # smallest key on the highest layer left
    links_after = {(l, int(i), int(t)) for l in range(after.n_levels) for r, i in enumerate(after.level_nodes[l])
                   for t in after.level_nbrs[l][r] if t != oracle.NONE}
    ds = set(dead)
    # the exports (dangling rows skipped) hold exactly the old live links between surviving nodes
    assert links_after == {(l, i, t) for l, i, t in links_before if i not in ds and t not in ds and l < after.n_levels}
    for l in range(after.n_levels):
        assert set(after.level_nodes[l].tolist()) == set(np.nonzero(alive_top >= l)[0].tolist())
    # degrees: one off per removed node the survivor was linked FROM (live or soft-deleted row of the removed node)
    dropped = sum(deg_before[(l, i)] - b.degree(i, l) for (l, i) in deg_before if i not in ds and l < after.n_levels)
    assert dropped >= len({(l, i, t) for l, i, t in links_before if t in ds and i not in ds and (l, t, i) in links_before})
    assert b.dangling_links() >= 0
    # the cleaned index still answers queries, and never with a removed node
    q = rng.random((16, 12), dtype=np.float32)
    ids, _, cnt, _ = after.knn_batch(q, 5, 30)
    assert (cnt == 5).all() and not np.isin(ids, dead).any()
    # inserting after a removal goes on from the cleaned graph
    b.insert(rng.random((50, 12), dtype=np.float32), oracle.random_levels(50, 6, 9))
    assert b.export().level_nodes[0].size == 1500 - len(dead) + 50


LITERAL_CASES = [
    # n, dim, metric, m, ef_c, extend, keep
    (140, 6, "L2", 3, 12, False, False),
    (140, 6, "L2", 3, 12, True, False),
    (120, 10, "Cosine", 4, 10, True, True),
    (120, 5, "IP", 3, 8, True, False),
    (100, 4, "L2", 2, 6, True, True),
]


@pytest.mark.parametrize("n,dim,metric,m,efc,extend,keep", LITERAL_CASES)
def test_index_construction_equals_a_literal_row_store(oracle, n, dim, metric, m, efc, extend, keep):
    """hnsw_put_vector / select_neighbours_heuristic / shrink_neighbour (hnsw.rs:155-538): the adjacency-list restatement in
    cozo_oracle.c against tests/literal_hnsw_store.py, which plays the same insertions on the reference's row model (one
    ordered map, get / put / del).  With extend_candidates this covers the self-link quirk: shrink selects the target itself,
    writes a link row onto the target's self row, and put_vector restores it -- a slot spent and a degree one above the
    number of link rows, no row of its own."""
    from tests.literal_hnsw_store import LiteralStore
    mid = {"L2": oracle.L2, "Cosine": oracle.COSINE, "IP": oracle.IP}[metric]
    x = util.vectors(n, dim, 5, "normal" if metric == "IP" else "uniform")
    levels = oracle.random_levels(n, m, 9)
    b = oracle.HnswBuilder(dim, mid, m, efc, extend_candidates=extend, keep_pruned_connections=keep)
    b.insert(x, levels)
    flat = b.export()
    st = LiteralStore(lambda a, c: oracle.distance(mid, a, c), m, efc, extend, keep)
    for i in range(n):
        st.put(x[i], int(levels[i]))
    assert st.entry() == flat.entry
    phantom = 0
    for lv in range(flat.n_levels):
        ids, tab = flat.level_nodes[lv], flat.level_nbrs[lv]
        assert [int(v) for v in ids] == [i for i in range(n) if levels[i] >= lv]
        for r, node in enumerate(ids):
            live = [int(t) for t in tab[r] if t != oracle.NONE]
            assert live == st.live_links(int(node), lv), (lv, int(node))
            assert b.degree(int(node), lv) == st.degree(int(node), lv), (lv, int(node))
            assert st.rows[(-lv, int(node), int(node))][1] is not None  # the self row is a self row again
            phantom += int(st.degree(int(node), lv)) - len(live)
    assert b.link_rows(include_ignored=True) - b.link_rows() == st.n_ignored()
    if extend:
        assert st.self_row_overwrites > 0 and phantom > 0
    else:
        assert st.self_row_overwrites == 0 and phantom == 0


@pytest.mark.parametrize("extend,keep", [(False, False), (True, False), (True, True)])
def test_rows_with_several_vectors_equal_a_literal_row_store(oracle, extend, keep):
    """Several indexed vectors per base row (hnsw.rs:694-706): hnsw_get_neighbours drops links inside a row (:609-610), so they
    are written and counted into the degrees but never searched, extended over, shrunk away or removed.  The oracle with
    orc_hnsw_set_row_of against the literal row store with the same rule."""
    from tests.literal_hnsw_store import LiteralStore
    n, dim, m, efc = 150, 5, 3, 10
    rng = np.random.default_rng(4)
    row_of = np.sort(rng.integers(0, n // 3, n)).astype(np.uint32)  # ~3 vectors per row, rows in key order
    base = rng.random((n // 3, dim), dtype=np.float32)
    x = (base[row_of] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)  # a row's vectors are near each other
    levels = oracle.random_levels(n, m, 2)
    b = oracle.HnswBuilder(dim, oracle.L2, m, efc, extend_candidates=extend, keep_pruned_connections=keep)
    b.set_row_of(row_of)
    b.insert(x, levels)
    flat = b.export()
    st = LiteralStore(lambda a, c: oracle.distance(oracle.L2, a, c), m, efc, extend, keep, row_of=row_of)
    for i in range(n):
        st.put(x[i], int(levels[i]))
    assert st.entry() == flat.entry
    hidden = 0
    for lv in range(flat.n_levels):
        ids, tab = flat.level_nodes[lv], flat.level_nbrs[lv]
        for r, node in enumerate(ids):
            live = [int(t) for t in tab[r] if t != oracle.NONE]
            assert live == st.live_links(int(node), lv), (lv, int(node))
            assert not any(row_of[t] == row_of[node] for t in live)
            assert b.degree(int(node), lv) == st.degree(int(node), lv), (lv, int(node))
            hidden += sum(1 for k, v in st.rows.items() if k[0] == -lv and k[1] == node and k[2] != node and not v[2]
                          and row_of[k[2]] == row_of[node])
    assert hidden > 0  # links inside a row were made, and stayed
    # a different index from the one that treats every vector as its own row
    plain = oracle.HnswBuilder(dim, oracle.L2, m, efc, extend_candidates=extend, keep_pruned_connections=keep)
    plain.insert(x, levels)
    assert not np.array_equal(plain.export().level_nbrs[0], flat.level_nbrs[0])


@pytest.mark.parametrize("extend", [False, True])
def test_removal_equals_a_literal_row_store(oracle, extend):
    """hnsw_remove_vec (hnsw.rs:754-868) restated in the oracle against the same deletions played on the literal row store:
    self rows and out rows at every layer, the reverse rows present or not, one off each neighbour's degree, the rows of
    OTHER nodes that pointed at the removed one left dangling; the entry point is whatever row comes first afterwards.  The
    reference panics when it removes a node that still holds such a dangling row (`.unwrap()` on the missing self row, :806-815):
    those removals are skipped in both models and counted."""
    import copy
    from tests.literal_hnsw_store import LiteralStore, ReferencePanic
    n, dim, m, efc = 160, 6, 3, 10
    x = util.vectors(n, dim, 8)
    levels = oracle.random_levels(n, m, 1)
    b = oracle.HnswBuilder(dim, oracle.L2, m, efc, extend_candidates=extend)
    b.insert(x, levels)
    st = LiteralStore(lambda a, c: oracle.distance(oracle.L2, a, c), m, efc, extend)
    for i in range(n):
        st.put(x[i], int(levels[i]))
    rng = np.random.default_rng(3)
    gone, panics = [], 0
    for v in rng.permutation(n)[:70]:
        trial = copy.deepcopy(st.rows)
        try:
            st.remove(int(v))
        except ReferencePanic:
            st.rows = trial
            panics += 1
            continue
        assert b.remove([int(v)]) == 1
        gone.append(int(v))
    assert len(gone) >= 40
    left = [i for i in range(n) if i not in set(gone)]
    flat = b.export()
    assert st.entry() == flat.entry
    assert b.dangling_links() == st.dangling()
    for lv in range(flat.n_levels):
        ids, tab = flat.level_nodes[lv], flat.level_nbrs[lv]
        assert [int(v) for v in ids] == [i for i in left if levels[i] >= lv]
        for r, node in enumerate(ids):
            assert [int(t) for t in tab[r] if t != oracle.NONE] == st.live_links(int(node), lv), (lv, int(node))
            assert b.degree(int(node), lv) == st.degree(int(node), lv), (lv, int(node))
    print(f"removed {len(gone)}, reference panics skipped {panics}, dangling rows {st.dangling()}")


@pytest.mark.parametrize("rows", [False, True])
def test_search_equals_the_literal_row_store(oracle, rows):
    """hnsw_knn (hnsw.rs:869-1012): the oracle's search over the exported flat tables against the literal walk over the row store
    -- descent with ef = 1, the bottom layer with ef, truncation to k only without a filter, radius, ascending rows -- also on
    an index whose base rows carry several vectors (the flat tables then hold no link inside a row, which is all a reader sees)."""
    from tests.literal_hnsw_store import LiteralStore
    n, dim, m, efc = 200, 7, 4, 12
    rng = np.random.default_rng(6)
    row_of = np.sort(rng.integers(0, n // 2, n)).astype(np.uint32) if rows else None
    x = util.vectors(n, dim, 15)
    levels = oracle.random_levels(n, m, 12)
    b = oracle.HnswBuilder(dim, oracle.L2, m, efc)
    if rows:
        b.set_row_of(row_of)
    b.insert(x, levels)
    flat = b.export()
    st = LiteralStore(lambda a, c: oracle.distance(oracle.L2, a, c), m, efc, row_of=row_of)
    for i in range(n):
        st.put(x[i], int(levels[i]))
    q = util.vectors(30, dim, 16)
    for k, ef, radius in ((5, 20, None), (8, 8, None), (6, 30, 0.35)):
        ids, dist, cnt, _ = flat.knn_batch(q, k, ef, radius=radius)
        for i in range(q.shape[0]):
            want = st.knn(q[i], k, ef, radius=radius)
            assert cnt[i] == len(want)
            assert [int(v) for v in ids[i][:cnt[i]]] == [v for v, _ in want]
            assert [float(d) for d in dist[i][:cnt[i]]] == [d for _, d in want]
    # with a filter the list is NOT cut to k before the filter runs (:943-947): k rows out of the ef found that pass
    keep = lambda v: v % 3 != 0
    ids, dist, cnt, _ = flat.knn_batch(q, 25, 25)
    for i in range(q.shape[0]):
        passed = [(int(v), float(d)) for v, d in zip(ids[i][:cnt[i]], dist[i][:cnt[i]]) if keep(int(v))][:4]
        assert passed == st.knn(q[i], 4, 25, accept=keep)


def test_distance_f64_arms(oracle):
    """VectorCache::dist's F64 arms (runtime/hnsw.rs:73-78, 86-95, 102-106): the known answers of runtime/tests.rs:693-694 in f64,
    and both summation orders (ndarray's unrolled_dot, the HIP kernels' tree) against numpy float64 on random vectors"""
    a, b = np.array([[1.0, 2.0]]), np.array([[2.0, 3.0]])
    pairs = np.array([[0, 0]], dtype=np.uint32)
    assert oracle.distance_pairs_f64(oracle.L2, b, a, pairs)[0] == 2.0
    assert abs(oracle.distance_pairs_f64(oracle.COSINE, a, a, pairs)[0]) <= 1e-15
    v = np.array([[0.6, 0.8]])
    assert abs(oracle.distance_pairs_f64(oracle.IP, v, v, pairs)[0]) <= 1e-15
    assert np.isnan(oracle.distance_pairs_f64(oracle.COSINE, np.zeros((1, 2)), a, pairs)[0])  # zero vector -> NaN
    rng = np.random.default_rng(5)
    for dim in (1, 2, 7, 8, 9, 128, 769, 1536):
        base, q = rng.standard_normal((40, dim)), rng.standard_normal((6, dim))
        pr = np.stack([rng.integers(0, 6, 200), rng.integers(0, 40, 200)], 1).astype(np.uint32)
        x, y = q[pr[:, 0]], base[pr[:, 1]]
        want = {oracle.L2: np.sum((x - y) ** 2, 1), oracle.COSINE: 1 - np.sum(x * y, 1) / np.sqrt(np.sum(x * x, 1) * np.sum(y * y, 1)),
                oracle.IP: 1 - np.sum(x * y, 1)}
        for metric in (oracle.L2, oracle.COSINE, oracle.IP):
            for mode in (oracle.DOT_NDARRAY, oracle.DOT_GPU):
                got = oracle.distance_pairs_f64(metric, base, q, pr, mode)
                scale = 1.0 + np.sum(np.abs(x * y), 1) + (np.sum((x - y) ** 2, 1) if metric == oracle.L2 else 0.0)
                assert np.max(np.abs(got - want[metric]) / scale) <= 1e-13, (dim, metric, mode)

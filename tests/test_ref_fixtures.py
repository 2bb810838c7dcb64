"""The oracle against the REFERENCE ITSELF: tests/golden/ref_fixtures.json is what the real cozo-core returned on the seeded
inputs of tests/golden/make_ref_inputs.py (produced by oracle/ref_fixtures/make_ref_fixtures.sh on a box with cargo).  The
image this repository is developed in has no Rust toolchain, so the file is absent there and these tests skip, saying so;
DESIGN.md section 3 therefore still reads "parity unpinned".  REF_FIXTURES=<path> points the tests at another file.

What is compared, and how strictly:
  distances      l2_dist / ip_dist / cos_dist f64 results == the oracle in ORC_DOT_NDARRAY order, bit for bit
  hnsw_knn       the reference's own index rows (`*tbl:idx{..}`) loaded into the oracle's flat layout, the oracle's search on
                 them == the reference's rows (keys and f64 distances), bit for bit -- the reference's random levels make the
                 BUILD unrepeatable (thread_rng, hnsw.rs:47-48), so the build is pinned through structure only: row counts per
                 level within m_max, symmetric stored distances
  PageRank       BOTH readings of graph 0.3.1's loop are restated (contributions refreshed after the sweep = Jacobi, or inside it =
                 in place; orc_pagerank_mode) and the test says which one the reference's rows equal bit for bit -- on a 3 000-node
                 graph (one chunk: deterministic either way), on a 40 000-node graph with default threads (three runs) and with
                 RAYON_NUM_THREADS=1 (tests/golden/ref_fixtures_pagerank_1thread.json), and at convergence (check_pagerank)
  CC / Dijkstra / ShortestPathBFS   group ids, f32 costs (as f64) and path lengths equal
  extend_candidates / rows with several vectors (round 3)   what the device build and the oracle ASSUME about the reference's
                 rows, as predictions on the reference-built index: every self row is a self row again after a shrink wrote a
                 link onto it (hash column not null), the degree sits 0 or 1 above the live link rows and somewhere 1; links
                 inside a base row exist in the store, are not among what the search walks, and the degree may count them;
                 and the oracle's search over what a reader sees of those rows == the reference's rows, bit for bit
"""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.environ.get("REF_FIXTURES") or os.path.join(HERE, "golden", "ref_fixtures.json")
needs_ref = pytest.mark.skipif(not os.path.exists(PATH), reason="no reference-generated fixtures (needs cargo: oracle/ref_fixtures/make_ref_fixtures.sh); parity stays unpinned")


@pytest.fixture(scope="module")
def ref():
    with open(PATH) as f:
        d = json.load(f)["results"]
    bad = [k for k, v in d.items() if v.get("ok") is not True]
    assert not bad, f"reference steps that failed: {bad}"
    return d


PATH_1T = os.environ.get("REF_FIXTURES_1THREAD") or os.path.join(HERE, "golden", "ref_fixtures_pagerank_1thread.json")


@pytest.fixture(scope="module")
def ref_one_thread():
    if not os.path.exists(PATH_1T):
        return None
    with open(PATH_1T) as f:
        return json.load(f)["results"]


@pytest.fixture(scope="module")
def inputs():
    spec = importlib.util.spec_from_file_location("make_ref_inputs", os.path.join(HERE, "golden", "make_ref_inputs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.dataset()


def _bits(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


def check_distances(ref, inputs, oracle):
    mod, d = inputs
    for dim in mod.DIMS:
        a, b = d["pairs"][dim]
        rows = ref[f"distances d={dim}"]["rows"]
        assert [r[0] for r in rows] == list(range(len(a)))
        pairs = np.stack([np.arange(len(a), dtype=np.uint32)] * 2, 1).copy()
        for col, metric in ((1, oracle.L2), (2, oracle.IP), (3, oracle.COSINE)):
            want = np.array([np.nan if r[col] is None else r[col] for r in rows], dtype=np.float64)
            got = oracle.distance_pairs(metric, b, a, pairs, oracle.DOT_NDARRAY)  # dist(query a[i], row b[i]); the metrics are symmetric
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(got), nan), (dim, metric)
            assert np.array_equal(_bits(got[~nan]), _bits(want[~nan])), (dim, metric, got[~nan][:3], want[~nan][:3])


def _flat_from_rows(oracle, rows, vectors, metric, m):
    """the reference's `tbl:idx` rows -> the flat layout (level l = layer -l; self rows and ignore_link rows are not links)"""
    by_level = {}
    for layer, fr_k, _ff, _fs, to_k, _tf, _ts, dist, ignore in rows:
        lvl = by_level.setdefault(-int(layer), {})
        lst = lvl.setdefault(int(fr_k), [])
        if int(fr_k) != int(to_k) and not ignore:
            lst.append((int(to_k), float(dist)))
    n_levels = max(by_level) + 1
    nodes, nbrs = [], []
    for l in range(n_levels):
        ids = sorted(by_level[l])
        width = max(1, max(len(by_level[l][i]) for i in ids))
        assert width <= (2 * m if l == 0 else m)
        tab = np.full((len(ids), width), oracle.NONE, dtype=np.uint32)
        for r, i in enumerate(ids):
            tos = sorted(t for t, _ in by_level[l][i])
            tab[r, :len(tos)] = tos
        nodes.append(np.array(ids, dtype=np.uint32))
        nbrs.append(tab)
    entry = int(nodes[-1][0])  # the first row of the index relation: smallest key on the top level (hnsw.rs:891-899)
    return oracle.FlatIndex(vectors, metric, nodes, nbrs, entry), by_level


def check_hnsw(ref, inputs, oracle):
    mod, d = inputs
    H = mod.HNSW
    for name, metric in (("L2", oracle.L2), ("Cosine", oracle.COSINE), ("IP", oracle.IP)):
        flat, by_level = _flat_from_rows(oracle, ref[f"hnsw index rows {name}"]["rows"], d["vectors"], metric, H["m"])
        assert sorted(by_level[0]) == list(range(H["n"]))
        # stored link distances are the metric's value for that pair, in the reference's arithmetic
        some = [(i, t, dist) for i in list(by_level[0])[:50] for t, dist in by_level[0][i]]
        pairs = np.array([[i, t] for i, t, _ in some], dtype=np.uint32)
        got = oracle.distance_pairs(metric, d["vectors"], d["vectors"], pairs, oracle.DOT_NDARRAY)
        assert np.array_equal(_bits(got), _bits([x for _, _, x in some])), name
        ids, dist, cnt, _ = flat.knn_batch(d["queries"], H["k"], H["ef"])
        want = {}
        for qi, k, dd in ref[f"hnsw knn {name}"]["rows"]:
            want.setdefault(int(qi), []).append((float(dd), int(k)))
        for qi in range(H["queries"]):
            w = sorted(want.get(qi, []))
            g = sorted((float(dist[qi, j]), int(ids[qi, j])) for j in range(cnt[qi]))
            assert [k for _, k in g] == [k for _, k in w], (name, qi)
            assert np.array_equal(_bits([x for x, _ in g]), _bits([x for x, _ in w])), (name, qi)


def _graph(oracle, d):
    from tests import util
    return util.graph_from_relation(oracle, d["frm"], d["to"]), util.graph_from_relation(oracle, d["frm"], d["to"], undirected=True), \
        util.graph_from_relation(oracle, d["frm"], d["to"], weights=d["w"])


PR_STEPS = (("pagerank defaults", "small", (0.85, 1e-4, 10)), ("pagerank theta=0.5 4 iterations", "small", (0.5, 0.0, 4)),
            ("pagerank big run 1", "big", (0.85, 1e-4, 10)), ("pagerank big run 2", "big", (0.85, 1e-4, 10)),
            ("pagerank big run 3", "big", (0.85, 1e-4, 10)), ("pagerank big converged", "big", (0.85, 1e-9, 200)))


def _pr_graph(oracle, d, which):
    from tests import util
    return util.graph_from_relation(oracle, d["frm"], d["to"]) if which == "small" else util.graph_from_relation(oracle, d["big_frm"], d["big_to"])


def _pr_want(rows, g):
    index_of = {int(v): i for i, v in enumerate(g["ind"])}
    assert len(rows) == g["n"]
    want = np.empty(g["n"], dtype=np.float64)
    for n, r in rows:
        want[index_of[int(n)]] = r
    return want


def pagerank_readings(oracle, g, args):
    """the four one-thread restatements of graph 0.3.1's loop: (contribution refresh: Jacobi | in place) x (error: f32 | f64 difference)"""
    out = {}
    for mode, mname in ((oracle.PR_JACOBI, "jacobi"), (oracle.PR_INPLACE, "inplace")):
        for ed in (False, True):
            s, it, err = oracle.pagerank_mode(g["n"], g["ioff"], g["isrc"], g["outdeg"], *args, mode=mode, err_f64_diff=ed)
            out[f"{mname}/{'f64' if ed else 'f32'}-diff"] = (s.astype(np.float64), it)
    return out


def check_pagerank(ref, inputs, oracle, one_thread=None):
    """Which reading of `graph::page_rank` is the reference's?  (VERDICT r3 weak #1b.)
    * small graph (3 000 nodes = ONE 16 384-node chunk = one rayon task = deterministic under either reading): the reference's
      rows must be bit-identical to exactly one family of restatements -- that DECIDES it, and the test fails loudly if neither is.
    * big graph (40 000 nodes, several chunks): under the Jacobi reading every run and the one-thread run are bit-identical to the
      Jacobi restatement; under the in-place reading only RAYON_NUM_THREADS=1 is deterministic (== the in-place restatement) and
      the default-thread runs differ from run to run -- reported, and then the only thread-count-independent statement left is the
      converged vector, which both restatements must reach within north_star's 1e-5.
    Returns the verdict; prints one line per step (pytest -s)."""
    _, d = inputs
    verdict = {}
    for src_name, src in (("default threads", ref), ("RAYON_NUM_THREADS=1", one_thread)):
        if src is None:
            continue
        for step, which, args in PR_STEPS:
            if step not in src:
                continue
            g = _pr_graph(oracle, d, which)
            want = _pr_want(src[step]["rows"], g)
            line = []
            for name, (got, _it) in pagerank_readings(oracle, g, args).items():
                rel = np.abs(got - want) / np.abs(want)
                kind = "bit-identical" if np.array_equal(got, want) else ("within 1e-5" if rel.max() <= 1e-5 else f"NEITHER (max rel {rel.max():.2e})")
                verdict[(src_name, step, name)] = kind
                line.append(f"{name}: {kind}")
            print(f"[{src_name}] {step}: " + "; ".join(line))
    # the deciding case
    for src_name in ("default threads", "RAYON_NUM_THREADS=1"):
        for step in ("pagerank defaults", "pagerank theta=0.5 4 iterations"):
            kinds = {k[2].split("/")[0]: v for k, v in verdict.items() if k[0] == src_name and k[1] == step and v == "bit-identical"}
            if any(k[0] == src_name and k[1] == step for k in verdict):
                assert kinds, f"{src_name} / {step}: the reference's scores equal NEITHER reading bit for bit: {[(k, v) for k, v in verdict.items() if k[1] == step]}"
    decided = {k[2].split("/")[0] for k, v in verdict.items() if k[1] == "pagerank defaults" and v == "bit-identical"}
    if decided:
        print("graph::page_rank refreshes contributions:", " / ".join(sorted(decided)))
    if "pagerank big converged" in ref:  # whatever the reading: the fixed point is the same within the tolerance
        for k, v in verdict.items():
            if k[1] == "pagerank big converged":
                assert v != "" and not v.startswith("NEITHER"), (k, v)
    return verdict


def _bfs_path(parent, start, goal):
    if goal != start and parent[goal] == 0xFFFFFFFF:
        return None
    path = [goal]
    while path[-1] != start:
        path.append(int(parent[path[-1]]))
        assert len(path) <= len(parent)
    return path[::-1]


def check_components_dijkstra_bfs(ref, inputs, oracle):
    _, d = inputs
    g, u, gw = _graph(oracle, d)
    index_of = {int(v): i for i, v in enumerate(g["ind"])}
    grp, _ = oracle.tarjan_groups(u["n"], u["ooff"], u["otgt"])
    rows = ref["connected components"]["rows"]
    assert len(rows) == g["n"]
    for n, gid in rows:
        assert int(grp[index_of[int(n)]]) == int(gid), n
    dist, _ = oracle.dijkstra(gw["n"], gw["ooff"], gw["otgt"], gw["ow"], index_of[d["dijkstra_start"]])
    seen = 0
    for s, t, c, p in ref["dijkstra"]["rows"]:
        assert int(s) == d["dijkstra_start"]
        got = float(dist[index_of[int(t)]])
        want = float("inf") if c is None else float(c)  # JSON has no inf: cozo prints it as null
        assert got == want, (t, got, want)
        assert (len(p) == 0) == np.isinf(want)
        seen += 1
    assert seen == g["n"]
    goals = np.array([index_of[x] for x in d["bfs_goals"]], dtype=np.uint32)
    start = index_of[d["bfs_start"]]
    parent = oracle.shortest_path_bfs(g["n"], g["ooff"], g["otgt"], start, goals)
    by_goal = {int(gv): p for _, gv, p in ref["shortest path bfs"]["rows"]}
    for gv, gi in zip(d["bfs_goals"], goals):
        want = by_goal[gv]
        got = _bfs_path(parent, start, int(gi))
        assert (want is None) == (got is None), gv
        if want is not None:  # FIFO order + sorted adjacency: the same path, node for node
            assert [index_of[int(x)] for x in want] == got, gv


def _flat_of_nodes(oracle, rows, node_of, vectors, metric, m):
    """ten-column `tbl:idx` rows -> the flat layout over nodes node_of[(k, sub)] (ids in key order); what a READER sees: no self
    rows, no soft-deleted rows, no link inside a base row (hnsw.rs:609-610).  Also: per (level, node) the self row's degree and
    hash, the live links a reader sees, the live links inside the base row."""
    info = {}
    for layer, fr_k, _ff, fr_s, to_k, _tf, to_s, dist, hsh, ignore in rows:
        fr, to = node_of[(int(fr_k), int(fr_s))], node_of[(int(to_k), int(to_s))]
        e = info.setdefault((-int(layer), fr), dict(degree=None, hash=None, seen=[], hidden=[]))
        if fr == to:
            e["degree"], e["hash"] = float(dist), hsh
        elif not ignore:
            (e["hidden"] if int(fr_k) == int(to_k) else e["seen"]).append(to)
    n_levels = max(l for l, _ in info) + 1
    nodes, nbrs = [], []
    for l in range(n_levels):
        ids = sorted(v for lv, v in info if lv == l)
        width = max(1, max(len(info[(l, v)]["seen"]) for v in ids))
        assert width <= (2 * m if l == 0 else m)
        tab = np.full((len(ids), width), oracle.NONE, dtype=np.uint32)
        for r, v in enumerate(ids):
            tos = sorted(info[(l, v)]["seen"])
            tab[r, :len(tos)] = tos
        nodes.append(np.array(ids, dtype=np.uint32))
        nbrs.append(tab)
    return oracle.FlatIndex(vectors, metric, nodes, nbrs, int(nodes[-1][0])), info


def check_extend_candidates(ref, inputs, oracle):
    mod, d = inputs
    E = mod.EXT
    node_of = {(i, -1): i for i in range(E["n"])}
    flat, info = _flat_of_nodes(oracle, ref["ext index rows"]["rows"], node_of, d["ext_vectors"], oracle.L2, E["m"])
    assert sorted(v for l, v in info if l == 0) == list(range(E["n"]))
    above = 0
    for (l, v), e in info.items():
        assert e["hash"] is not None, (l, v)  # the self row was put back (hnsw.rs:352-357) after the shrink wrote onto it (:413-433)
        extra = e["degree"] - len(e["seen"])
        assert extra in (0.0, 1.0), (l, v, e["degree"], len(e["seen"]))  # the target selected itself: a slot, no row
        above += int(extra)
    assert above > 0
    ids, dist, cnt, _ = flat.knn_batch(d["ext_queries"], E["k"], E["ef"])
    want = {}
    for qi, k, dd in ref["ext knn"]["rows"]:
        want.setdefault(int(qi), []).append((float(dd), int(k)))
    for qi in range(E["queries"]):
        g = sorted((float(dist[qi, j]), int(ids[qi, j])) for j in range(cnt[qi]))
        assert g == sorted(want.get(qi, [])), qi


def _row_nodes(d):
    keys = [(k, s) for k, vs in enumerate(d["rows_vectors"]) for s in range(len(vs))]
    return {ks: i for i, ks in enumerate(keys)}, keys, np.stack([d["rows_vectors"][k][s] for k, s in keys]).astype(np.float32)


def check_rows_with_several_vectors(ref, inputs, oracle):
    mod, d = inputs
    R = mod.ROWS
    node_of, keys, vectors = _row_nodes(d)
    flat, info = _flat_of_nodes(oracle, ref["rows index rows"]["rows"], node_of, vectors, oracle.L2, R["m"])
    assert sorted(v for l, v in info if l == 0) == list(range(len(keys)))
    hidden = 0
    for (l, v), e in info.items():
        assert e["hash"] is not None
        hidden += len(e["hidden"])
        # counted at insertion like any link (:281-357); a later shrink re-counts only what hnsw_get_neighbours returns (:609-610)
        assert len(e["seen"]) <= e["degree"] <= len(e["seen"]) + len(e["hidden"]), (l, v, e)
    assert hidden > 0  # links inside a base row ARE written
    ids, dist, cnt, _ = flat.knn_batch(d["rows_queries"], R["k"], R["ef"])
    want = {}
    for qi, k, sub, dd in ref["rows knn"]["rows"]:
        want.setdefault(int(qi), []).append((float(dd), node_of[(int(k), int(sub))]))
    for qi in range(R["queries"]):
        g = sorted((float(dist[qi, j]), int(ids[qi, j])) for j in range(cnt[qi]))
        assert g == sorted(want.get(qi, [])), qi


@needs_ref
def test_distances_bit_for_bit(ref, inputs, oracle):
    check_distances(ref, inputs, oracle)


@needs_ref
def test_hnsw_knn_on_the_reference_built_index(ref, inputs, oracle):
    check_hnsw(ref, inputs, oracle)


@needs_ref
def test_pagerank_which_reading_of_graph_page_rank(ref, ref_one_thread, inputs, oracle):
    check_pagerank(ref, inputs, oracle, ref_one_thread)


@needs_ref
def test_components_dijkstra_and_bfs(ref, inputs, oracle):
    check_components_dijkstra_bfs(ref, inputs, oracle)


@needs_ref
def test_extend_candidates_on_the_reference_built_index(ref, inputs, oracle):
    check_extend_candidates(ref, inputs, oracle)


@needs_ref
def test_rows_with_several_vectors_on_the_reference_built_index(ref, inputs, oracle):
    check_rows_with_several_vectors(ref, inputs, oracle)


def test_the_checks_themselves_on_rows_the_oracle_produced(inputs, oracle):
    """No reference output exists in this image, so the four checks above would never run here and could rot.  This feeds them
    a stand-in `results` object built from the ORACLE's own outputs in the row shapes cozo returns (headers as in
    make_ref_inputs.steps): it proves nothing about parity, only that the plumbing a maintainer will run is sound."""
    mod, d = inputs
    from tests import util
    fake = {}
    for dim in mod.DIMS:
        a, b = d["pairs"][dim]
        pairs = np.stack([np.arange(len(a), dtype=np.uint32)] * 2, 1).copy()
        cols = [oracle.distance_pairs(m, b, a, pairs, oracle.DOT_NDARRAY) for m in (oracle.L2, oracle.IP, oracle.COSINE)]
        fake[f"distances d={dim}"] = dict(ok=True, rows=[[i] + [None if np.isnan(c[i]) else float(c[i]) for c in cols] for i in range(len(a))])
    H = mod.HNSW
    for name, metric in (("L2", oracle.L2), ("Cosine", oracle.COSINE), ("IP", oracle.IP)):
        _, flat = util.build_index(oracle, d["vectors"], metric, H["m"], H["ef_construction"])
        rows = []
        for l in range(flat.n_levels):
            for r, i in enumerate(flat.level_nodes[l]):
                tos = [int(t) for t in flat.level_nbrs[l][r] if t != oracle.NONE]
                rows.append([-l, int(i), 1, -1, int(i), 1, -1, float(len(tos)), False])  # the self row carries the degree
                if tos:
                    pr = np.array([[int(i), t] for t in tos], dtype=np.uint32)
                    dd = oracle.distance_pairs(metric, d["vectors"], d["vectors"], pr, oracle.DOT_NDARRAY)
                    rows += [[-l, int(i), 1, -1, t, 1, -1, float(x), False] for t, x in zip(tos, dd)]
        fake[f"hnsw index rows {name}"] = dict(ok=True, rows=rows)
        ids, dist, cnt, _ = flat.knn_batch(d["queries"], H["k"], H["ef"])
        fake[f"hnsw knn {name}"] = dict(ok=True, rows=[[qi, int(ids[qi, j]), float(dist[qi, j])] for qi in range(H["queries"]) for j in range(cnt[qi])])
    g, u, gw = _graph(oracle, d)
    fake_inplace = {}
    for step, which, args in PR_STEPS:
        gg = _pr_graph(oracle, d, which)
        s, _, _ = oracle.pagerank(gg["n"], gg["ioff"], gg["isrc"], gg["outdeg"], *args)
        fake[step] = dict(ok=True, rows=[[int(gg["ind"][i]), float(s[i])] for i in range(gg["n"])])
        s2, _, _ = oracle.pagerank_mode(gg["n"], gg["ioff"], gg["isrc"], gg["outdeg"], *args, mode=oracle.PR_INPLACE, err_f64_diff=True)
        fake_inplace[step] = dict(ok=True, rows=[[int(gg["ind"][i]), float(s2[i])] for i in range(gg["n"])])
    grp, _ = oracle.tarjan_groups(u["n"], u["ooff"], u["otgt"])
    fake["connected components"] = dict(ok=True, rows=[[int(g["ind"][i]), int(grp[i])] for i in range(g["n"])])
    index_of = {int(v): i for i, v in enumerate(g["ind"])}
    dist, par = oracle.dijkstra(gw["n"], gw["ooff"], gw["otgt"], gw["ow"], index_of[d["dijkstra_start"]])
    fake["dijkstra"] = dict(ok=True, rows=[[d["dijkstra_start"], int(g["ind"][i]), None if np.isinf(dist[i]) else float(dist[i]),
                                            [] if np.isinf(dist[i]) else [0]] for i in range(g["n"])])
    goals = np.array([index_of[x] for x in d["bfs_goals"]], dtype=np.uint32)
    start = index_of[d["bfs_start"]]
    parent = oracle.shortest_path_bfs(g["n"], g["ooff"], g["otgt"], start, goals)
    fake["shortest path bfs"] = dict(ok=True, rows=[[d["bfs_start"], gv, (lambda p: None if p is None else [int(g["ind"][x]) for x in p])(_bfs_path(parent, start, int(gi)))]
                                                    for gv, gi in zip(d["bfs_goals"], goals)])
    # the two round-3 paths: rows of the literal row store (tests/literal_hnsw_store.py), which keeps ALL rows like the reference
    from tests.literal_hnsw_store import LiteralStore
    dist_l2 = lambda a, c: oracle.distance(oracle.L2, a, c, oracle.DOT_NDARRAY)
    E, R = mod.EXT, mod.ROWS
    st = LiteralStore(dist_l2, E["m"], E["ef_construction"], extend_candidates=True)
    for v, lv in zip(d["ext_vectors"], oracle.random_levels(E["n"], E["m"], 11)):
        st.put(v, int(lv))
    fake["ext index rows"] = dict(ok=True, rows=[[la, fr, 1, -1, to, 1, -1, val[0], None if val[1] is None else "h", val[2]]
                                                 for (la, fr, to), val in sorted(st.rows.items())])
    fake["ext knn"] = dict(ok=True, rows=[[qi, node, dd] for qi, q in enumerate(d["ext_queries"]) for node, dd in st.knn(q, E["k"], E["ef"])])
    node_of, keys, vectors = _row_nodes(d)
    row_of = np.array([k for k, _ in keys], dtype=np.uint32)
    st = LiteralStore(dist_l2, R["m"], R["ef_construction"], row_of=row_of)
    for v, lv in zip(vectors, oracle.random_levels(len(keys), R["m"], 12)):
        st.put(v, int(lv))
    fake["rows index rows"] = dict(ok=True, rows=[[la, keys[fr][0], 1, keys[fr][1], keys[to][0], 1, keys[to][1], val[0],
                                                   None if val[1] is None else "h", val[2]] for (la, fr, to), val in sorted(st.rows.items())])
    fake["rows knn"] = dict(ok=True, rows=[[qi, keys[node][0], keys[node][1], dd] for qi, q in enumerate(d["rows_queries"])
                                           for node, dd in st.knn(q, R["k"], R["ef"])])
    check_extend_candidates(fake, inputs, oracle)
    check_rows_with_several_vectors(fake, inputs, oracle)
    check_distances(fake, inputs, oracle)
    check_hnsw(fake, inputs, oracle)
    v = check_pagerank(fake, inputs, oracle)
    assert v[("default threads", "pagerank defaults", "jacobi/f32-diff")] == "bit-identical"
    assert v[("default threads", "pagerank defaults", "inplace/f32-diff")].startswith("NEITHER")  # 10 sweeps: the readings are > 1e-5 apart
    assert not v[("default threads", "pagerank big converged", "inplace/f64-diff")].startswith("NEITHER")  # ... and meet at the fixed point
    v = check_pagerank(fake_inplace, inputs, oracle, fake_inplace)  # a reference that refreshes in place would be told apart
    assert v[("RAYON_NUM_THREADS=1", "pagerank defaults", "inplace/f64-diff")] == "bit-identical"
    assert v[("default threads", "pagerank defaults", "jacobi/f32-diff")].startswith("NEITHER")
    check_components_dijkstra_bfs(fake, inputs, oracle)

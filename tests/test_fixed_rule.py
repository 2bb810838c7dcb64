"""Rule-level parity: cozo_amd/fixed_rule.py (the host mirror of cozo-core's FixedRule surface) against literal,
DataValue-level restatements of the reference rules written here (small inputs), which read like the
reference's own `run` bodies.  Every case runs twice: on CPU with the oracle standing in for the device library
(host logic: options, id mapping, CSR build, emission) and, marked `gpu`, through the C ABI on the device."""
import heapq
import json
import math
import os
from collections import deque

import numpy as np
import pytest

from cozo_amd import fixed_rule as FR
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))

BACKENDS = [pytest.param("oracle", id="host-logic"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


@pytest.fixture(params=BACKENDS)
def registry(request, monkeypatch, oracle):
    if request.param == "oracle":
        util.OracleGraphBackend(oracle).install(monkeypatch)
    else:
        request.getfixturevalue("gpu_lib")
    return FR.FixedRuleRegistry()


def rel(rows, bindings=None, arity=None):
    return FR.FixedRuleInputRelation(rows, bindings, arity)


# ---- literal restatements (DataValue level) -------------------------------------------------------------------
def _adj(edge_rows):
    """edges.prefix_iter(node): rows with that first column in key order"""
    rows = sorted({FR._canon(tuple(r)): tuple(r) for r in edge_rows}.values(), key=FR._tuple_key)
    adj = {}
    for r in rows:
        adj.setdefault(FR._canon(r[0]), []).append(r)
    return adj


def ref_shortest_path_bfs(edge_rows, starts, ends):
    """shortest_path_bfs.rs:45-110, line by line"""
    adj = _adj(edge_rows)
    starts = [t[0] for t in sorted({FR._canon(s): (s,) for s in starts}.values(), key=FR._tuple_key)]
    ends = sorted({FR._canon(e): e for e in ends}.values(), key=FR.sort_key)
    out = []
    for s in starts:
        pending = {FR._canon(e) for e in ends}
        visited = {FR._canon(s)}
        back = {}
        q = deque([s])
        while q:
            cand = q.pop()
            for edge in adj.get(FR._canon(cand), ()):
                to = edge[1]
                if FR._canon(to) in visited:
                    continue
                visited.add(FR._canon(to))
                back[FR._canon(to)] = cand
                pending.discard(FR._canon(to))
                if not pending:
                    break
                q.appendleft(to)
        for e in ends:
            if FR._canon(e) in back:
                route, cur = [], e
                while FR._canon(cur) != FR._canon(s):
                    route.append(cur)
                    cur = back[FR._canon(cur)]
                route.append(s)
                out.append((s, e, route[::-1]))
            else:
                out.append((s, e, None))
    return out


def ref_bfs(edge_rows, node_rows, start_rows, condition, limit, skip_query_nodes):
    """bfs.rs:35-113"""
    adj = _adj(edge_rows)
    nodes = _adj(node_rows)
    starts = sorted({FR._canon(tuple(r)): tuple(r) for r in start_rows}.values(), key=FR._tuple_key)
    visited, back, found = set(), {}, []
    done = False
    for st in starts:
        s = st[0]
        if FR._canon(s) in visited:
            continue
        visited.add(FR._canon(s))
        q = deque([s])
        while q and not done:
            cand = q.pop()
            for edge in adj.get(FR._canon(cand), ()):
                to = edge[1]
                if FR._canon(to) in visited:
                    continue
                visited.add(FR._canon(to))
                back[FR._canon(to)] = cand
                if skip_query_nodes:
                    tup = (to,)
                else:
                    lst = nodes.get(FR._canon(to))
                    if not lst:
                        raise FR.NodeNotFoundError(cand)
                    tup = lst[0]
                if condition(tup):
                    found.append((s, to))
                    if len(found) >= limit:
                        done = True
                        break
                q.appendleft(to)
        if done:
            break
    out = []
    for s, e in found:
        route, cur = [], e
        while FR._canon(cur) != FR._canon(s):
            route.append(cur)
            cur = back[FR._canon(cur)]
        route.append(s)
        out.append((s, e, route[::-1]))
    return out


def _first_appearance(edge_rows):
    rows = sorted({FR._canon(tuple(r)): tuple(r) for r in edge_rows}.values(), key=FR._tuple_key)
    idx, inv = [], {}
    for r in rows:
        for v in (r[0], r[1]):
            if FR._canon(v) not in inv:
                inv[FR._canon(v)] = len(idx)
                idx.append(v)
    return rows, idx, inv


def ref_connected_components(edge_rows, node_rows=None):
    """strongly_connected_components.rs:50-75 with strong = false: group id = rank of the component by its
    smallest first-appearance index (what Tarjan over ascending roots yields on a symmetric graph)"""
    rows, idx, inv = _first_appearance(edge_rows)
    par = list(range(len(idx)))

    def find(x):
        while par[x] != x:
            par[x] = par[par[x]]
            x = par[x]
        return x

    for r in rows:
        a, b = find(inv[FR._canon(r[0])]), find(inv[FR._canon(r[1])])
        if a != b:
            par[max(a, b)] = min(a, b)
    roots = sorted({find(i) for i in range(len(idx))})
    gid = {r: i for i, r in enumerate(roots)}
    out = [(idx[i], gid[find(i)]) for i in range(len(idx))]
    counter = len(roots)
    if node_rows is not None:
        seen = set(inv)
        for t in sorted({FR._canon(tuple(r)): tuple(r) for r in node_rows}.values(), key=FR._tuple_key):
            if FR._canon(t[0]) not in seen:
                seen.add(FR._canon(t[0]))
                out.append((t[0], counter))
                counter += 1
    return out


def ref_dijkstra_costs(edge_rows, undirected):
    """f32 Dijkstra over the weighted relation: {(start value, target value): cost}; also the edge weights"""
    rows, idx, inv = _first_appearance(edge_rows)
    adj = {}
    for r in rows:
        w = np.float32(1.0 if len(r) < 3 else r[2])
        a, b = inv[FR._canon(r[0])], inv[FR._canon(r[1])]
        adj.setdefault(a, []).append((b, w))
        if undirected:
            adj.setdefault(b, []).append((a, w))

    def run(s):
        dist = [np.float32(np.inf)] * len(idx)
        dist[s] = np.float32(0)
        pq = [(np.float32(0), s)]
        while pq:
            c, u = heapq.heappop(pq)
            if c > dist[u]:
                continue
            for v, w in adj.get(u, ()):
                nc = np.float32(c + w)
                if nc < dist[v]:
                    dist[v] = nc
                    heapq.heappush(pq, (nc, v))
        return dist

    return idx, inv, adj, run


def ref_pagerank(edge_rows, undirected, theta, epsilon, iterations):
    """graph::page_rank restated in numpy f32 (sequential sums in sorted in-neighbour order, Jacobi)"""
    rows, idx, inv = _first_appearance(edge_rows)
    n = len(idx)
    pairs = []
    for r in rows:
        a, b = inv[FR._canon(r[0])], inv[FR._canon(r[1])]
        pairs.append((a, b))
        if undirected:
            pairs.append((b, a))
    ins = [[] for _ in range(n)]
    outdeg = np.zeros(n, dtype=np.uint32)
    for a, b in pairs:
        ins[b].append(a)
        outdeg[a] += 1
    for l in ins:
        l.sort()
    d = np.float32(theta)
    init = np.float32(1.0) / np.float32(n)
    base = (np.float32(1.0) - d) / np.float32(n)
    score = np.full(n, init, dtype=np.float32)
    with np.errstate(divide="ignore"):
        contrib = (score / outdeg.astype(np.float32)).astype(np.float32)
    tol = float(np.float32(epsilon))
    for _ in range(iterations):
        new = np.empty(n, dtype=np.float32)
        err = 0.0
        for u in range(n):
            s = np.float32(0)
            for v in ins[u]:
                s = np.float32(s + contrib[v])
            new[u] = np.float32(base + np.float32(d * s))
            err += abs(float(np.float32(new[u] - score[u])))
        score = new
        with np.errstate(divide="ignore"):
            contrib = (score / outdeg.astype(np.float32)).astype(np.float32)
        if err < tol:
            break
    return [(idx[i], float(score[i])) for i in range(n)]


def rowset(rows):
    return sorted(((FR._canon(tuple(r)), r) for r in rows), key=lambda x: FR._tuple_key(x[1]))


def assert_same_rows(got, want):
    g = [c for c, _ in rowset(got)]
    w = [c for c, _ in rowset(want)]
    assert g == w


# ---- inputs ----------------------------------------------------------------------------------------------------
def love_graph():
    with open(os.path.join(HERE, "golden", "love_graph.json")) as f:
        return json.load(f)


def str_graph(n, e, seed):
    rng = np.random.default_rng(seed)
    names = [f"n{int(i):03d}" for i in rng.permutation(n)]
    return [(names[a], names[b]) for a, b in zip(rng.integers(0, n, e), rng.integers(0, n, e))]


# ---- tests -----------------------------------------------------------------------------------------------------
def test_shortest_path_bfs_love_graph_golden(registry):
    """the reference's own pinned result (shortest_path_bfs.rs:124-174): alice -> bob has 3 hops, george is Null"""
    g = love_graph()
    rows = registry.run("ShortestPathBFSGpu", [rel(g["edges"]), rel([["alice"]]), rel([["bob"], ["george"]])])
    by_end = {r[1]: r[2] for r in rows}
    for ex in g["expect"]:
        if ex["len"] is None:
            assert by_end[ex["to"]] is None
        else:
            assert len(by_end[ex["to"]]) == ex["len"]
    assert by_end["bob"] == ["alice", "eve", "bob"]
    assert_same_rows(rows, ref_shortest_path_bfs(g["edges"], ["alice"], ["bob", "george"]))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_shortest_path_bfs_random_string_keys(registry, seed):
    edges = str_graph(60, 150, seed)
    rng = np.random.default_rng(seed + 100)
    nodes = sorted({v for e in edges for v in e})
    starts = [nodes[i] for i in rng.integers(0, len(nodes), 4)] + ["not-a-node"]
    ends = [nodes[i] for i in rng.integers(0, len(nodes), 6)] + ["also-missing", starts[0]]
    rows = registry.run("ShortestPathBFSGpu", [rel(edges), rel([[s] for s in starts]), rel([[e] for e in ends])])
    assert_same_rows(rows, ref_shortest_path_bfs(edges, starts, ends))


def test_shortest_path_bfs_mixed_value_types(registry):
    # Null < Bool < Num < Str ordering decides the neighbour scan order, hence which of two equal-length paths wins
    edges = [(1, "b"), (1, 2.5), (1, None), ("b", "z"), (2.5, "z"), (None, "z"), (1, True), (True, "z")]
    rows = registry.run("ShortestPathBFSGpu", [rel(edges), rel([[1]]), rel([["z"]])])
    assert rows == [(1, "z", [1, None, "z"])]
    assert_same_rows(rows, ref_shortest_path_bfs(edges, [1], ["z"]))


def test_shortest_path_bfs_arity_errors(registry):
    with pytest.raises(FR.InputRelationArityError):
        registry.run("ShortestPathBFSGpu", [rel([(1,)]), rel([[1]]), rel([[1]])])
    with pytest.raises(FR.FixedRuleInputNotFoundError):
        registry.run("ShortestPathBFSGpu", [rel([(1, 2)]), rel([[1]])])


@pytest.mark.parametrize("limit", [1, 3, 1000])
def test_bfs_rule(registry, limit):
    edges = str_graph(50, 120, 11)
    nodes = sorted({v for e in edges for v in e})
    node_rows = [(v, int(v[1:]) % 7) for v in nodes]
    cond = lambda t: t[1] == 3  # noqa: E731
    starts = [(nodes[5],), (nodes[17],), (nodes[5],), (nodes[40],)]
    rows = registry.run("BFSGpu", [rel(edges), rel(node_rows, ["id", "tag"]), rel(starts)],
                        {"limit": limit, "condition": cond})
    assert_same_rows(rows, ref_bfs(edges, node_rows, starts, cond, limit, False))
    assert len(rows) <= limit


def test_bfs_rule_defaults_and_errors(registry):
    edges = [(1, 2), (2, 3), (3, 4), (1, 5)]
    node_rows = [(i,) for i in range(1, 5)]  # node 5 is missing from `nodes`
    with pytest.raises(FR.FixedRuleOptionNotFoundError):
        registry.run("BFSGpu", [rel(edges), rel(node_rows)], {})
    with pytest.raises(FR.WrongFixedRuleOptionError):
        registry.run("BFSGpu", [rel(edges), rel(node_rows)], {"condition": lambda t: True, "limit": 0})
    # starting nodes default to `nodes` (bfs.rs:37); discovery order 2, 5 -> the missing node raises before 4 is seen
    with pytest.raises(FR.NodeNotFoundError):
        registry.run("BFSGpu", [rel(edges), rel(node_rows), rel([(1,)])], {"condition": lambda t: t[0] == 4})
    cond = lambda t: t[0] == 4  # noqa: E731
    cond.only_node_id = True  # the condition binds only the id column: `nodes` is not consulted (bfs.rs:40)
    rows = registry.run("BFSGpu", [rel(edges), rel(node_rows), rel([(1,)])], {"condition": cond})
    assert rows == [(1, 4, [1, 2, 3, 4])]


@pytest.mark.parametrize("seed", [5, 6])
def test_connected_components_rule(registry, seed):
    edges = str_graph(80, 70, seed)
    extra = [("lonely-1",), ("lonely-0",), (edges[0][0],)]
    rows = registry.run("ConnectedComponentsGpu", [rel(edges), rel(extra)])
    assert_same_rows(rows, ref_connected_components(edges, extra))
    rows = registry.run("ConnectedComponentsGpu", [rel(edges)])
    assert_same_rows(rows, ref_connected_components(edges))


def test_connected_components_int_keys_fast_path(registry):
    frm, to = util.random_relation(500, 400, 9)
    edges = list(zip(frm.tolist(), to.tolist()))
    rows = registry.run("ConnectedComponentsGpu", [rel(edges)])
    assert_same_rows(rows, ref_connected_components(edges))


def ref_clustering_coefficients(edge_rows):
    """triangles.rs:25-110 at the DataValue level: first-appearance ids, symmetrised multi-adjacency"""
    rows = sorted({FR._canon(tuple(r)): tuple(r) for r in edge_rows}.values(), key=FR._tuple_key)
    ids, vals = {}, []
    adj = {}
    for f, t in ((r[0], r[1]) for r in rows):
        for v in (f, t):
            if FR._canon(v) not in ids:
                ids[FR._canon(v)] = len(vals)
                vals.append(v)
        a, b = ids[FR._canon(f)], ids[FR._canon(t)]
        adj.setdefault(a, []).append(b)
        adj.setdefault(b, []).append(a)
    out = []
    for v in range(len(vals)):
        edges = sorted(adj.get(v, []))
        d = len(edges)
        if d < 2:
            out.append((vals[v], 0.0, 0, d))
            continue
        t = sum(1 for s in edges for e in edges if s > e and e in adj.get(s, []))
        out.append((vals[v], 2.0 * t / (d * (d - 1.0)), t, d))
    return out


def ref_degree_centrality(edge_rows, node_rows=()):
    """degree_centrality.rs:24-76"""
    counter, vals = {}, {}
    for r in sorted({FR._canon(tuple(r)): tuple(r) for r in edge_rows}.values(), key=FR._tuple_key):
        for pos, v in ((1, r[0]), (2, r[1])):
            vals.setdefault(FR._canon(v), v)
            c = counter.setdefault(FR._canon(v), [0, 0, 0])
            c[0] += 1
            c[pos] += 1
    for r in node_rows:
        vals.setdefault(FR._canon(r[0]), r[0])
        counter.setdefault(FR._canon(r[0]), [0, 0, 0])
    return [(vals[k], c[0], c[1], c[2]) for k, c in counter.items()]


@pytest.mark.parametrize("seed", [11, 12])
def test_clustering_coefficients_rule(registry, seed):
    edges = str_graph(60, 200, seed)
    edges += [(b, a) for a, b in edges[:20]] + [(edges[0][0], edges[0][0])]  # bidirectional pairs, a self loop
    rows = registry.run("ClusteringCoefficientsGpu", [rel(edges)])
    assert_same_rows(rows, ref_clustering_coefficients(edges))
    assert registry.run("ClusteringCoefficientsGpu", [rel([])]) == []


def ref_closeness_centrality(edge_rows, undirected):
    """all_pairs_shortest_path.rs:97-176 at the DataValue level, f32 arithmetic in the reference's order"""
    rows = sorted({FR._canon(tuple(r)): tuple(r) for r in edge_rows}.values(), key=FR._tuple_key)
    ids, vals, adj = {}, [], {}
    for r in rows:
        for v in (r[0], r[1]):
            if FR._canon(v) not in ids:
                ids[FR._canon(v)] = len(vals)
                vals.append(v)
        a, b, w = ids[FR._canon(r[0])], ids[FR._canon(r[1])], np.float32(r[2] if len(r) > 2 else 1.0)
        adj.setdefault(a, []).append((b, w))
        if undirected:
            adj.setdefault(b, []).append((a, w))
    n = len(vals)
    out = []
    for start in range(n):
        dist = [np.float32(np.inf)] * n
        dist[start] = np.float32(0.0)
        pq = [(np.float32(0.0), start)]
        while pq:
            cost, node = heapq.heappop(pq)
            if cost > dist[node]:
                continue
            for tgt, w in adj.get(node, ()):
                nxt = np.float32(cost + w)
                if nxt < dist[tgt]:
                    dist[tgt] = nxt
                    heapq.heappush(pq, (nxt, tgt))
        total, nc = np.float32(0.0), np.float32(0.0)
        for d in dist:
            if np.isfinite(d):
                total = np.float32(total + d)
                nc = np.float32(nc + np.float32(1.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.float32(np.float32(np.float32(nc * nc) / total) / np.float32(n - 1))
        out.append((vals[start], float(c)))
    return out


@pytest.mark.parametrize("undirected", [False, True])
def test_closeness_centrality_rule(registry, undirected):
    rng = np.random.default_rng(17)
    names = [f"n{i:02d}" for i in range(40)]
    edges = sorted({(names[a], names[b]) for a, b in rng.integers(0, 40, (150, 2)) if a != b})
    edges = [(a, b, float(np.float32(rng.integers(1, 64) / 8.0))) for a, b in edges]
    rows = registry.run("ClosenessCentralityGpu", [rel(edges)], {"undirected": undirected})
    want = ref_closeness_centrality(edges, undirected)
    got = {FR._canon(r[0]): r[1] for r in rows}
    assert len(rows) == len(want)
    for node, c in want:
        g = got[FR._canon(node)]
        assert g == c or (math.isnan(g) and math.isnan(c))


def test_degree_centrality_rule(registry):
    edges = str_graph(40, 90, 3)
    extra = [("nobody",), (edges[0][0],)]
    assert_same_rows(registry.run("DegreeCentralityGpu", [rel(edges), rel(extra)]), ref_degree_centrality(edges, extra))
    assert_same_rows(registry.run("DegreeCentralityGpu", [rel(edges)]), ref_degree_centrality(edges))
    with pytest.raises(FR.InputRelationArityError):
        registry.run("DegreeCentralityGpu", [rel([("a",)])])


@pytest.mark.parametrize("undirected", [False, True])
def test_dijkstra_rule(registry, undirected):
    rng = np.random.default_rng(21)
    base = str_graph(40, 110, 21)
    edges = list({(a, b): (a, b, float(np.float32(rng.random() * 10))) for a, b in base}.values())
    nodes = sorted({v for e in edges for v in e[:2]})
    starts = [(nodes[3],), (nodes[9],), ("ghost",)]
    rows = registry.run("ShortestPathDijkstraGpu", [rel(edges), rel(starts)], {"undirected": undirected})
    idx, inv, adj, run = ref_dijkstra_costs(edges, undirected)
    assert len(rows) == 2 * len(idx)
    wmap = {}
    for a, lst in adj.items():
        for b, w in lst:
            wmap[(a, b)] = min(w, wmap.get((a, b), np.float32(np.inf)))
    for s, t, cost, path in rows:
        d = run(inv[FR._canon(s)])
        assert np.float32(cost) == d[inv[FR._canon(t)]]  # f32 costs are bit-exact
        if math.isfinite(cost):
            assert path[0] == s and path[-1] == t
            c = np.float32(0)
            for a, b in zip(path, path[1:]):
                c = np.float32(c + wmap[(inv[FR._canon(a)], inv[FR._canon(b)])])
            assert c == np.float32(cost)  # the emitted path is a tight one
        else:
            assert path == []
    # termination set: only those targets are emitted
    goals = [(nodes[0],), (nodes[1],), ("ghost",)]
    rows2 = registry.run("ShortestPathDijkstraGpu", [rel(edges), rel(starts[:1]), rel(goals)], {"undirected": undirected})
    assert sorted(r[1] for r in rows2) == sorted([nodes[0], nodes[1]])
    full = {(r[0], r[1]): r[2] for r in rows}
    for s, t, cost, _ in rows2:
        assert full[(s, t)] == cost


def test_dijkstra_rule_weight_errors_and_defaults(registry):
    for bad in ("x", float("nan"), float("inf"), -1.0, None):
        with pytest.raises(FR.BadEdgeWeightError):
            registry.run("ShortestPathDijkstraGpu", [rel([(1, 2, bad)]), rel([(1,)])])
    rows = registry.run("ShortestPathDijkstraGpu", [rel([(1, 2)]), rel([(1,)])])  # default weight 1.0
    assert rows == [(1, 1, 0.0, [1]), (1, 2, 1.0, [1, 2])]
    # keep_ties without a termination relation is the plain run (shortest_path_dijkstra.rs:73-86 never reaches
    # dijkstra_keep_ties then); with one it is tests/test_zz_tie_rules.py
    assert registry.run("ShortestPathDijkstraGpu", [rel([(1, 2, 1.0)]), rel([(1,)])], {"keep_ties": True}) == rows


@pytest.mark.parametrize("undirected", [False, True])
def test_pagerank_rule(registry, undirected):
    edges = str_graph(70, 300, 31)
    rows = registry.run("PageRankGpu", [rel(edges)], {"undirected": undirected, "iterations": 7})
    want = ref_pagerank(edges, undirected, 0.85, 0.0001, 7)
    got = {r[0]: r[1] for r in rows}
    assert set(got) == {w[0] for w in want}
    for node, score in want:
        if math.isfinite(score):
            assert abs(got[node] - score) <= 1e-5 * abs(score)  # north_star tolerance; in practice bit-identical
            assert got[node] == score
        else:
            assert got[node] == score or (math.isnan(got[node]) and math.isnan(score))


def test_pagerank_rule_options(registry):
    assert registry.run("PageRankGpu", [rel([])]) == []  # pagerank.rs:43-45
    for opts in ({"theta": 1.5}, {"epsilon": -0.1}, {"iterations": 0}, {"undirected": 1}, {"theta": "x"}):
        with pytest.raises(FR.WrongFixedRuleOptionError):
            registry.run("PageRankGpu", [rel([(1, 2)])], opts)
    with pytest.raises(FR.NotAnEdgeError):
        registry.run("PageRankGpu", [rel([(1,)])])


def test_registry_rules():
    reg = FR.FixedRuleRegistry()
    with pytest.raises(FR.FixedRuleNameConflict):
        reg.register_fixed_rule("PageRank", FR.PageRank())  # built-in name (runtime/db.rs:766-771)
    with pytest.raises(FR.FixedRuleNameConflict):
        reg.register_fixed_rule("PageRankGpu", FR.PageRank())
    reg.register_fixed_rule("MyRank", FR.PageRank())
    assert reg.get("MyRank").arity({}, ()) == 2
    assert reg.unregister_fixed_rule("MyRank") and not reg.unregister_fixed_rule("MyRank")
    with pytest.raises(FR.FixedRuleNameConflict):
        reg.unregister_fixed_rule("PageRank")


def test_as_directed_graph_first_appearance_and_sorted_csr(oracle):
    """ids: first appearance, source before destination, rows in key order (mod.rs:163-180); adjacency ascending
    with parallel edges kept.  Int fast path == generic path == the oracle's restatement."""
    frm, to = util.random_relation(300, 900, 3, self_loops=True)
    rows = [(int(a), int(b), i) for i, (a, b) in enumerate(zip(frm, to))] + [(int(frm[0]), int(to[0]), -1)]
    r = rel(rows)
    g, idx, inv = r.as_directed_graph(False)
    rr = [t for t in r.iter()]
    fi, ti, ind = oracle.assign_ids(np.array([t[0] for t in rr]), np.array([t[1] for t in rr]))
    assert idx == ind.tolist()
    ooff, otgt = oracle.build_csr(len(ind), fi, ti)
    ioff, isrc = oracle.build_csr(len(ind), ti, fi)
    assert np.array_equal(g.out_offsets, ooff) and np.array_equal(g.out_targets, otgt)
    assert np.array_equal(g.in_offsets, ioff) and np.array_equal(g.in_sources, isrc)
    # generic (non-int) path gives the same ids
    r2 = rel([(f"k{a:05d}", f"k{b:05d}", c) for a, b, c in rows])
    fi2, ti2, idx2, _ = r2._assign_first_appearance(*r2._edge_columns())
    rr2 = [t for t in r2.iter()]
    fi3, ti3, idx3, _ = r._assign_first_appearance([int(t[0][1:]) for t in rr2], [int(t[1][1:]) for t in rr2])
    assert np.array_equal(fi2, fi3) and np.array_equal(ti2, ti3) and [int(s[1:]) for s in idx2] == idx3
    gu, _, _ = r.as_directed_graph(True)
    uoff, utgt = oracle.build_csr(len(ind), fi, ti, undirected=True)
    assert np.array_equal(gu.out_offsets, uoff) and np.array_equal(gu.out_targets, utgt)


def test_poison_is_honoured(registry):
    p = FR.Poison()
    p.kill()
    with pytest.raises(Exception) as ei:
        registry.run("ShortestPathBFSGpu", [rel([(1, 2)]), rel([[1]]), rel([[2]])], poison=p)
    assert "kill" in str(ei.value).lower() or "cancel" in str(ei.value).lower()


def test_custom_rule_plumbing_like_the_reference():
    """runtime/tests.rs:529-577 `test_custom_rules`: a user rule registered next to the GPU rules, fed by rel[] <- [[1,2,3,4],
    [5,6,7,8]] with mult: 100, answers [[1000], [2600]] -- the boundary conformance case the reference itself pins."""
    class Custom(FR.FixedRule):
        def arity(self, options, rule_head):
            return 1

        def run(self, payload, out, poison):
            rel_ = payload.get_input(0)
            mult = payload.integer_option("mult", 2)
            for row in rel_.iter():
                out.put((sum(c if type(c) is int else 0 for c in row) * mult,))

    reg = FR.FixedRuleRegistry()
    reg.register_fixed_rule("SumCols", Custom())
    assert reg.run("SumCols", [rel([[1, 2, 3, 4], [5, 6, 7, 8]])], {"mult": 100}) == [(1000,), (2600,)]
    assert reg.run("SumCols", [rel([[1, 2, 3, 4], [5, 6, 7, 8]])]) == [(20,), (52,)]  # the default, mult = 2
    with pytest.raises(FR.FixedRuleNameConflict):
        reg.register_fixed_rule("SumCols", Custom())  # runtime/db.rs:769-774
    with pytest.raises(FR.FixedRuleNameConflict):
        reg.register_fixed_rule("PageRank", Custom())  # a built-in name
    with pytest.raises(FR.FixedRuleNameConflict):
        reg.unregister_fixed_rule("PageRank")  # :780-782
    assert reg.unregister_fixed_rule("SumCols") and not reg.unregister_fixed_rule("SumCols")


def ref_label_propagation_in_python(edge_rows, undirected, max_iter, colour_of):
    """label_propagation.rs:56-109 word for word in Python (BTreeMap -> dict kept in insertion order of ascending... no: a dict
    keyed by label, summed in adjacency order), with the node order (colour classes ascending, ids ascending inside) and the
    tie-break (smallest label) the GPU rule fixes"""
    rows = sorted({FR._canon(tuple(r)): tuple(r) for r in edge_rows}.values(), key=FR._tuple_key)  # a relation is a sorted set
    ids, vals = {}, []
    for r in rows:
        for v in (r[0], r[1]):
            if v not in ids:
                ids[v] = len(vals)
                vals.append(v)
    triples = []
    for r in rows:
        w = np.float32(r[2]) if len(r) > 2 else np.float32(1.0)
        triples.append((ids[r[0]], ids[r[1]], w))
        if undirected:
            triples.append((ids[r[1]], ids[r[0]], w))
    triples.sort(key=lambda t: (t[0], t[1]))  # CsrLayout::Sorted: by target inside a node, stable for parallel edges
    n = len(vals)
    adj = [[] for _ in range(n)]
    for a, b, w in triples:
        adj[a].append((b, w))
    colour = colour_of(n, adj)
    order = sorted(range(n), key=lambda v: (colour[v], v))
    labels = list(range(n))
    for _ in range(max_iter):
        changed = False
        for node in order:
            scores = {}
            for t, w in adj[node]:
                lab = labels[t]
                scores[lab] = np.float32(scores.get(lab, np.float32(0.0)) + w)
            if not scores:
                continue
            best = max(scores.values())
            new = min(lab for lab, sc in scores.items() if sc == best)
            if new != labels[node]:
                changed, labels[node] = True, new
        if not changed:
            break
    return sorted(((labels[i], vals[i]) for i in range(n)), key=lambda r: (r[0], r[1]))


def _python_colouring(n, adj):
    def key(v):
        x = v
        x ^= x >> 16
        x = (x * 0x85EBCA6B) & 0xFFFFFFFF
        x ^= x >> 13
        x = (x * 0xC2B2AE35) & 0xFFFFFFFF
        x ^= x >> 16
        return (x << 32) | v
    nb = [set() for _ in range(n)]
    for a in range(n):
        for b, _ in adj[a]:
            if a != b:
                nb[a].add(b)
                nb[b].add(a)
    colour, r = [None] * n, 0
    while any(c is None for c in colour):
        take = [v for v in range(n) if colour[v] is None and all(colour[u] is not None or key(u) < key(v) for u in nb[v])]
        for v in take:
            colour[v] = r
        r += 1
    return colour


@pytest.mark.parametrize("undirected", [False, True])
@pytest.mark.parametrize("weighted", [False, True])
def test_label_propagation_rule(registry, undirected, weighted):
    """LabelPropagationGpu == the reference's loop run with the fixed node order and tie-break (a Python restatement of
    label_propagation.rs:56-109 and of the colouring; the oracle's C restatement is checked against the same in
    tests/test_oracle.py).  Three planted communities with a few cross edges; weights in quarter steps make score ties."""
    rng = np.random.default_rng(11 + undirected + 2 * weighted)
    edges = []
    for c in range(3):
        members = list(range(c * 12, c * 12 + 12))
        for _ in range(40):
            a, b = rng.choice(members, 2, replace=False)
            edges.append((int(a), int(b)))
    for _ in range(6):
        edges.append((int(rng.integers(0, 36)), int(rng.integers(0, 36))))
    rows = [(f"n{a}", f"n{b}", float(rng.integers(1, 9)) / 4) if weighted else (f"n{a}", f"n{b}") for a, b in edges]
    got = registry.run("LabelPropagationGpu", [rel(rows)], {"undirected": undirected, "max_iter": 10})
    want = ref_label_propagation_in_python(rows, undirected, 10, _python_colouring)
    assert got == want
    assert len({r[0] for r in got}) < len(got) / 3  # labels did spread: far fewer labels than nodes
    assert registry.run("LabelPropagationGpu", [rel([])]) == []
    # max_iter is honoured: one iteration changes less than ten
    one = registry.run("LabelPropagationGpu", [rel(rows)], {"undirected": undirected, "max_iter": 1})
    assert one == ref_label_propagation_in_python(rows, undirected, 1, _python_colouring)
    with pytest.raises(Exception):
        registry.run("LabelPropagationGpu", [rel(rows)], {"max_iter": 0})

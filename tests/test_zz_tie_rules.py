"""The two rules built on `dijkstra_keep_ties` (shortest_path_dijkstra.rs:341-450).  BetweennessCentralityGpu (SURVEY section 8
f3): device SSSP from every node + Brandes accumulation on the tight-edge DAG, against the oracle's literal restatement of the
reference (enumeration of all shortest paths).  ShortestPathDijkstraGpu with keep_ties: every shortest path is a row.  Host logic
on CPU with the oracle standing in for cz_sssp, and, marked gpu, through the C ABI on the device.  (In a file of its own so that
it runs after the established device tests.)"""
import numpy as np
import pytest

from cozo_amd import fixed_rule as FR
from tests import util

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


@pytest.mark.parametrize("undirected", [False, True])
@pytest.mark.parametrize("kind", ["float-weights", "small-int-weights", "unit-weights"])
def test_betweenness_centrality_rule(registry, oracle, undirected, kind):
    """BetweennessCentralityGpu (device SSSP from every node + Brandes accumulation on the tight-edge DAG) against the oracle's
    literal restatement of the reference (dijkstra_keep_ties + enumeration of ALL shortest paths, f32 accumulation in the
    reference's order): within 1e-5 relative.  Integer and unit weights make ties -- several shortest paths per pair."""
    rng = np.random.default_rng({"float-weights": 5, "small-int-weights": 6, "unit-weights": 7}[kind])
    n, e = (45, 220) if kind == "float-weights" else (28, 90)
    names = [f"n{i:02d}" for i in range(n)]
    pairs = sorted({(names[a], names[b]) for a, b in rng.integers(0, n, (e, 2)) if a != b})
    if kind == "float-weights":
        edges = [(a, b, float(np.float32(rng.random() * 9 + 0.5))) for a, b in pairs]
    elif kind == "small-int-weights":
        edges = [(a, b, int(rng.integers(1, 4))) for a, b in pairs]
    else:
        edges = pairs  # two columns: every weight is 1.0 (fixed_rule/mod.rs:226)
    rows = registry.run("BetweennessCentralityGpu", [rel(edges)], {"undirected": undirected})
    r = rel(edges)
    graph, indices, _ = r.as_directed_weighted_graph(undirected, False)
    want = oracle.betweenness(graph.n, graph.out_offsets, graph.out_targets, graph.out_weights)
    got = {FR._canon(node): c for node, c in rows}
    assert len(rows) == graph.n and float(want.max()) > 0.0
    if kind != "float-weights":
        assert np.any(np.abs(want - np.round(want)) > 1e-3)  # fractional shares: ties were really exercised
    for i, node in enumerate(indices):
        assert got[FR._canon(node)] == pytest.approx(float(want[i]), rel=1e-5, abs=1e-6), node


def test_betweenness_centrality_rule_edges(registry):
    assert registry.run("BetweennessCentralityGpu", [rel([])]) == []
    assert registry.run("BetweennessCentralityGpu", [rel([("a", "b", 1.0), ("b", "c", 1.0)])]) == [("a", 0.0), ("b", 1.0), ("c", 0.0)]
    assert registry.run("BetweennessCentralityGpu", [rel([("a", "b"), ("b", "c")])], {"undirected": True}) == [("a", 0.0), ("b", 2.0), ("c", 0.0)]
    with pytest.raises(FR.FixedRuleError):
        registry.run("BetweennessCentralityGpu", [rel([("a", "b", 0.0)])])
    with pytest.raises(FR.BadEdgeWeightError):
        registry.run("BetweennessCentralityGpu", [rel([("a", "b", -1.0)])])


# ---- ShortestPathDijkstra with keep_ties (shortest_path_dijkstra.rs:341-450): every shortest path is a row ------------------
def ref_dijkstra_keep_ties(edge_rows, undirected, start, goals):
    """dijkstra_keep_ties restated literally over values: f32 costs, back pointers cleared on a strict improvement and
    appended on an equal cost, every path through them collected; returns {(target, cost, tuple(path))}"""
    import heapq
    adj = {}
    for r in edge_rows:
        w = np.float32(1.0 if len(r) < 3 else r[2])
        adj.setdefault(r[0], []).append((r[1], w))
        if undirected:
            adj.setdefault(r[1], []).append((r[0], w))
    dist, back = {start: np.float32(0)}, {}
    pq = [(np.float32(0), start)]
    while pq:
        cost, node = heapq.heappop(pq)
        if cost > dist.get(node, np.float32(np.inf)):
            continue
        for nxt, w in adj.get(node, ()):
            nc = np.float32(cost + w)
            cur = dist.get(nxt, np.float32(np.inf))
            if nc < cur:
                dist[nxt] = nc
                back[nxt] = [node]
                heapq.heappush(pq, (nc, nxt))
            elif nc == cur:
                back[nxt].append(node)
    out = set()
    for t in goals:
        if t not in dist:
            out.add((t, float("inf"), ()))
            continue

        def collect(chain):
            for nxt in back.get(chain[-1], ()):
                if nxt == start:
                    out.add((t, float(dist[t]), tuple(reversed(chain + [nxt]))))
                else:
                    collect(chain + [nxt])
        collect([t])
    return out


@pytest.mark.parametrize("undirected", [False, True])
def test_dijkstra_keep_ties_rule(registry, undirected):
    rng = np.random.default_rng(31)
    names = [f"n{i:02d}" for i in range(30)]
    pairs = sorted({(names[a], names[b]) for a, b in rng.integers(0, 30, (110, 2)) if a != b})
    edges = [(a, b, int(rng.integers(1, 3))) for a, b in pairs]  # weights 1 or 2: plenty of equal-cost paths
    nodes = sorted({v for e in edges for v in e[:2]})
    starts, goals = [nodes[0], nodes[4]], nodes[::2]
    rows = registry.run("ShortestPathDijkstraGpu", [rel(edges), rel([(s,) for s in starts]), rel([(g,) for g in goals] + [("ghost",)])],
                        {"undirected": undirected, "keep_ties": True})
    want = set()
    for s in starts:
        for t, cost, path in ref_dijkstra_keep_ties(edges, undirected, s, goals):
            want.add((s, t, cost, path))
    got = {(s, t, c, tuple(p)) for s, t, c, p in rows}
    assert got == want
    by_pair = {}
    for s, t, c, p in got:
        by_pair.setdefault((s, t), []).append(p)
    assert max(len(v) for v in by_pair.values()) >= 2  # ties were really exercised
    assert (starts[0], starts[0]) not in by_pair       # the start as its own target collects nothing (:397-430)
    # one start and one goal (the `single` branch, :73-80) gives the same rows for that pair
    one = registry.run("ShortestPathDijkstraGpu", [rel(edges), rel([(starts[0],)]), rel([(goals[1],)])],
                       {"undirected": undirected, "keep_ties": True})
    assert {(s, t, c, tuple(p)) for s, t, c, p in one} == {r for r in got if r[0] == starts[0] and r[1] == goals[1]}


def test_betweenness_on_air_routes_against_networkx(registry, request):
    """An independent cross-check on the reference's own fixture (cozo-core/tests/air_routes.rs data): the routes among the
    220 busiest airports, weight = distance in miles (integers: f32 and f64 agree on every tie), BetweennessCentralityGpu against
    networkx's weighted betweenness (Brandes, float64, unnormalised, endpoints excluded) -- the same quantity the reference's
    path enumeration sums up."""
    import os
    nx = pytest.importorskip("networkx")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "air_routes.npz"))
    codes = [str(c) for c in z["codes"]]
    busiest = np.argsort(-np.bincount(z["fr"], minlength=len(codes)), kind="stable")[:220]  # the 220 airports with most routes out
    keep = {codes[i] for i in busiest}
    routes = [(codes[a], codes[b], float(d)) for a, b, d in zip(z["fr"], z["to"], z["dist"]) if codes[a] in keep and codes[b] in keep]
    assert len(routes) > 5000 and all(r[2] > 0 and r[2] == int(r[2]) for r in routes)
    rows = registry.run("BetweennessCentralityGpu", [rel(routes)])
    g = nx.DiGraph()
    g.add_weighted_edges_from(routes)
    want = nx.betweenness_centrality(g, normalized=False, weight="weight", endpoints=False)
    got = dict(rows)
    assert set(got) == set(want) and max(want.values()) > 100
    # the device sums in f64; the host-logic stand-in is the oracle's literal f32 enumeration (the reference's arithmetic)
    tol = 1e-9 if "gpu" in request.node.callspec.id else 1e-5
    for node, c in want.items():
        assert got[node] == pytest.approx(c, rel=tol, abs=tol), node


def test_clustering_coefficients_on_air_routes_against_networkx(registry):
    """ClusteringCoefficientsGpu on the reference's fixture with every unordered airport pair kept once (the rule symmetrises its
    input; a pair stored in both directions would count double, triangles.rs:36) == networkx's clustering / triangles / degree on
    the simple undirected graph"""
    import os
    nx = pytest.importorskip("networkx")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "air_routes.npz"))
    codes = [str(c) for c in z["codes"]]
    pairs = sorted({tuple(sorted((codes[a], codes[b]))) for a, b in zip(z["fr"], z["to"]) if a != b})
    rows = registry.run("ClusteringCoefficientsGpu", [rel(pairs)])
    g = nx.Graph()
    g.add_edges_from(pairs)
    cc, tri, deg = nx.clustering(g), nx.triangles(g), dict(g.degree())
    assert len(rows) == g.number_of_nodes() > 3000 and sum(tri.values()) > 100_000
    for node, c, t, d in rows:
        assert (t, d) == (tri[node], deg[node]) and c == pytest.approx(cc[node], rel=1e-12, abs=1e-15), node


def test_closeness_centrality_on_air_routes_against_scipy(registry):
    """ClosenessCentralityGpu on the 220 busiest airports of the reference's fixture against scipy's Dijkstra (float64;
    distances are integer miles, every partial sum stays below 2^24, so the reference's f32 epilogue is exact and the
    comparison is bit for bit): nc * nc / total / (n - 1), all_pairs_shortest_path.rs:118-122"""
    import os
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "air_routes.npz"))
    codes = [str(c) for c in z["codes"]]
    busiest = np.argsort(-np.bincount(z["fr"], minlength=len(codes)), kind="stable")[:220]
    keep = {codes[i] for i in busiest}
    routes = [(codes[a], codes[b], float(d)) for a, b, d in zip(z["fr"], z["to"], z["dist"]) if codes[a] in keep and codes[b] in keep]
    rows = registry.run("ClosenessCentralityGpu", [rel(routes)])
    nodes = sorted({r[0] for r in routes} | {r[1] for r in routes})
    pos = {c: i for i, c in enumerate(nodes)}
    n = len(nodes)
    m = csr_matrix(([r[2] for r in routes], ([pos[r[0]] for r in routes], [pos[r[1]] for r in routes])), shape=(n, n))
    d = dijkstra(m, directed=True)
    got = dict(rows)
    assert len(rows) == n
    for c, i in pos.items():
        fin = d[i][np.isfinite(d[i])]
        assert fin.sum() < 2 ** 24
        want = np.float32(np.float32(np.float32(fin.size) * np.float32(fin.size)) / np.float32(fin.sum())) / np.float32(n - 1)
        assert got[c] == float(want), c

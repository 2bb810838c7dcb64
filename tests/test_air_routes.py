"""The reference's own graph fixture (cozo-core/tests/air_routes.rs: `route{fr, to => dist}`, `airport{code}`)
through the rule-level mirror, against expectations computed by scipy.sparse.csgraph / float64 numpy
(tests/golden/make_air_routes_golden.py): independent of oracle/ and of the kernels, so the CPU leg pins the
oracle and the GPU leg pins the device path on the same data."""
import json
import math
import os

import numpy as np
import pytest

from cozo_amd import fixed_rule as FR
from tests import util

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


@pytest.fixture(scope="module")
def air():
    z = np.load(os.path.join(G, "air_routes.npz"))
    codes = [str(c) for c in z["codes"]]
    route = [(codes[a], codes[b], float(d)) for a, b, d in zip(z["fr"], z["to"], z["dist"])]
    exp = json.load(open(os.path.join(G, "air_routes_expect.json")))
    ez = np.load(os.path.join(G, "air_routes_expect.npz"))
    return dict(codes=codes, pos={c: i for i, c in enumerate(codes)}, route=FR.FixedRuleInputRelation(route),
                airport=FR.FixedRuleInputRelation([(c,) for c in codes], ["code"]), exp=exp, ez=ez, n_routes=len(route))


@pytest.fixture(params=BACKENDS)
def registry(request, monkeypatch, oracle):
    if request.param == "oracle":
        util.OracleGraphBackend(oracle).install(monkeypatch)
    else:
        request.getfixturevalue("gpu_lib")
    return FR.FixedRuleRegistry()


def test_route_count(air):
    assert air["n_routes"] == air["exp"]["routes"] == 50637  # air_routes.rs:208


def test_bfs_pek_lhr(air, registry):
    """air_routes.rs:212-236"""
    cond = lambda t: t[0] == "LHR"  # noqa: E731
    rows = registry.run("BFSGpu", [air["route"], air["airport"], FR.FixedRuleInputRelation([("PEK",)])], {"condition": cond})
    assert len(rows) == 1
    s, e, path = rows[0]
    assert (s, e, path[0], path[-1]) == ("PEK", "LHR", "PEK", "LHR")
    assert len(path) == air["exp"]["bfs"]["hops"] + 1


def test_shortest_path_bfs_hops(air, registry):
    hops = air["exp"]["bfs_hops_from_PEK"]
    rows = registry.run("ShortestPathBFSGpu", [air["route"], FR.FixedRuleInputRelation([("PEK",)]),
                                               FR.FixedRuleInputRelation([(c,) for c in hops])])
    got = {r[1]: r[2] for r in rows}
    for c, h in hops.items():
        if c == "PEK":
            assert got[c] is None or got[c][0] == "PEK"  # the start is only in the backtrace via a cycle
        elif h is None:
            assert got[c] is None
        else:
            assert len(got[c]) == h + 1 and got[c][0] == "PEK" and got[c][-1] == c
    # every hop of an emitted path is a route
    routes = {(t[0], t[1]) for t in air["route"].iter()}
    for p in got.values():
        if p:
            assert all((a, b) in routes for a, b in zip(p, p[1:]))


def test_shortest_path_bfs_all_hops(air, registry):
    """hop counts to EVERY airport equal scipy's unweighted shortest paths"""
    hp = air["ez"]["hops_pek"]
    rows = registry.run("ShortestPathBFSGpu", [air["route"], FR.FixedRuleInputRelation([("PEK",)]), air["airport"]])
    assert len(rows) == len(air["codes"])
    for s, e, p in rows:
        h = hp[air["pos"][e]]
        if e == "PEK":
            continue
        assert (p is None) == (not np.isfinite(h))
        if p is not None:
            assert len(p) == int(h) + 1


def test_connected_components(air, registry):
    """air_routes.rs:254-267"""
    rows = registry.run("ConnectedComponentsGpu", [air["route"], air["airport"]])
    assert len(rows) == len(air["codes"])
    lab = air["ez"]["cc_label"]
    in_graph = air["ez"]["in_graph"]
    got = {c: g for c, g in rows}
    ng = air["exp"]["cc"]["components_among_route_nodes"]
    a = np.array([got[c] for c in air["codes"]])
    # same partition on the route nodes
    pairs = set(zip(lab[in_graph].tolist(), a[in_graph].tolist()))
    assert len(pairs) == ng and len({p[0] for p in pairs}) == ng and len({p[1] for p in pairs}) == ng
    assert sorted({p[1] for p in pairs}) == list(range(ng))
    assert np.bincount(a[in_graph]).max() == air["exp"]["cc"]["largest"]
    # airports without routes: fresh ids after the components, in key order (strongly_connected_components.rs:61-74)
    lonely = [c for c in air["codes"] if not in_graph[air["pos"][c]]]
    assert [got[c] for c in sorted(lonely)] == list(range(ng, ng + len(lonely)))
    # the component of the first scanned row is group 0
    first = next(air["route"].iter())
    assert got[first[0]] == 0


def test_dijkstra_from_jfk(air, registry):
    """air_routes.rs:300-316 (JFK -> KUL) and every other target"""
    rows = registry.run("ShortestPathDijkstraGpu", [air["route"], FR.FixedRuleInputRelation([("JFK",)])])
    want = air["ez"]["dijkstra_jfk"]
    in_graph = air["ez"]["in_graph"]
    assert len(rows) == int(in_graph.sum())
    dist = {(t[0], t[1]): t[2] for t in air["route"].iter()}
    tot, reach = 0.0, 0
    for s, t, cost, path in rows:
        w = want[air["pos"][t]]
        if np.isfinite(w):
            assert cost == w  # integer miles: exact in f32
            assert path[0] == "JFK" and path[-1] == t
            assert sum(dist[(a, b)] for a, b in zip(path, path[1:])) == cost
            tot += cost
            reach += 1
        else:
            assert math.isinf(cost) and path == []
    assert reach == air["exp"]["dijkstra"]["reachable"] and tot == air["exp"]["dijkstra"]["sum_finite"]
    for c, w in air["exp"]["dijkstra"]["costs"].items():
        got = [r for r in rows if r[1] == c][0][2]
        assert got == w
    rows = registry.run("ShortestPathDijkstraGpu", [air["route"], FR.FixedRuleInputRelation([("JFK",)]),
                                                    FR.FixedRuleInputRelation([("KUL",)])])
    assert len(rows) == 1 and rows[0][:3] == ("JFK", "KUL", air["exp"]["dijkstra"]["costs"]["KUL"])


def test_pagerank_vs_float64(air, registry):
    rows = registry.run("PageRankGpu", [air["route"]], {"iterations": 10, "epsilon": 0.0})
    want = dict(zip([air["codes"][i] for i in air["ez"]["pagerank_nodes"]], air["ez"]["pagerank_f64"]))
    assert len(rows) == len(want)
    worst = max(abs(score - want[c]) / want[c] for c, score in rows)
    assert worst <= 1e-5, worst  # north_star tolerance, against an independent float64 implementation
    top = sorted(rows, key=lambda r: -r[1])[:10]
    assert [c for c, _ in top] == [c for c, _ in air["exp"]["pagerank"]["top10"]]


def test_degree_centrality_against_the_references_own_asserts(air):
    """PINNED BY THE REFERENCE: cozo-core/tests/air_routes.rs asserts these route counts on this very fixture
    (most_out_routes :475-505, most_routes :539-566, airport_with_one_route :570-586, airports_by_route_number
    :783-799).  DegreeCentrality (algos/degree_centrality.rs:24-76) computes the same counts per node."""
    ref = json.load(open(os.path.join(G, "air_routes_reference_asserts.json")))
    edges = FR.FixedRuleInputRelation([(t[0], t[1]) for t in air["route"].iter()])
    rows = FR.FixedRuleRegistry().run("DegreeCentralityGpu", [edges, air["airport"]])
    assert len(rows) == air["exp"]["airports"]  # airports without routes appear with zeros (:47-56)
    out_deg = {r[0]: r[2] for r in rows}
    total = {r[0]: r[1] for r in rows}
    assert all(r[1] == r[2] + r[3] for r in rows)
    top_out = sorted(((c, n) for c, n in out_deg.items() if n > 180), key=lambda x: (-x[1], x[0]))
    assert [list(x) for x in top_out] == ref["most_out_routes_gt_180"]
    top_tot = sorted(((c, n) for c, n in total.items() if n > 400), key=lambda x: (-x[1], x[0]))
    assert [list(x) for x in top_tot] == ref["most_routes_gt_400"]
    assert sum(1 for n in out_deg.values() if n == 1) == ref["airports_with_exactly_one_out_route"]
    assert sorted(c for c, n in out_deg.items() if n == 106) == ref["airports_with_106_out_routes"]
    for code, n in ref["routes_per_airport"]:
        assert out_deg[code] == n
    hist = {}
    for n in out_deg.values():
        hist[n] = hist.get(n, 0) + 1
    assert [[n, hist[n]] for n in sorted(hist)[:10]] == ref["group_count_by_out_first_10"]
    assert abs(sum(out_deg.values()) / len(out_deg) - ref["mean_out_routes_over_all_airports"]) <= 1e-8

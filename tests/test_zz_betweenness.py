"""BetweennessCentralityGpu: device SSSP from every node + Brandes accumulation on the tight-edge DAG,
against the oracle's literal restatement of the reference (dijkstra_keep_ties + enumeration of all shortest paths).  Host logic
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

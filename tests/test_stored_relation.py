"""The fixed rules run off the STORED BYTES of their edge relation (cozo_amd/stored_relation.py -> libcozo_ingest) and
off the decoded tuples (the reference's route): identical rows.  Host logic on CPU with the oracle standing in for the
device library, and, marked gpu, through the C ABI on the device."""
import numpy as np
import pytest

from cozo_amd import codec
from cozo_amd import fixed_rule as FR
from cozo_amd.stored_relation import StoredInputRelation
from tests import util

BACKENDS = [pytest.param("oracle", id="host-logic"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


@pytest.fixture(params=BACKENDS)
def registry(request, monkeypatch, oracle):
    from cozo_amd import build as B
    B.build_ingest()
    if request.param == "oracle":
        util.OracleGraphBackend(oracle).install(monkeypatch)
    else:
        request.getfixturevalue("gpu_lib")
    return FR.FixedRuleRegistry()


def _edges(seed, n, e, strings=False, weights=False):
    rng = np.random.default_rng(seed)
    rows = set()
    while len(rows) < e:
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if a != b:
            rows.add((a, b))
    name = (lambda i: f"n{i:03d}") if strings else (lambda i: i * 7 - 50)
    if weights:
        return [(name(a), name(b), float(rng.integers(1, 40)) / 4) for a, b in sorted(rows)]
    return [(name(a), name(b)) for a, b in sorted(rows)]


def both(tuples, n_key_cols):
    """the same relation twice: as decoded tuples and as stored bytes (with fewer key columns than columns, later rows
    replace earlier ones that share the key -- the decoded side is read back from the store)"""
    rows = codec.StoredRows.from_tuples(3, tuples, n_key_cols)
    return FR.FixedRuleInputRelation([tuple(t) for t in rows.tuples()]), StoredInputRelation(rows)


def rel(rows):
    return FR.FixedRuleInputRelation(rows)


@pytest.mark.parametrize("strings", [False, True])
@pytest.mark.parametrize("n_key_cols", [2, 1])
def test_rules_read_stored_bytes(registry, strings, n_key_cols):
    tuples = _edges(1, 60, 240, strings)
    plain, stored = both(tuples, n_key_cols)
    nodes = sorted({t[0] for t in plain.iter()} | {t[1] for t in plain.iter()})
    starts, ends = [[nodes[0]], [nodes[7]]], [[nodes[3]], [nodes[11]], [nodes[-1]]]
    for name, extra, opts in [("PageRankGpu", [], {}), ("PageRankGpu", [], {"undirected": True, "iterations": 4}),
                              ("ConnectedComponentsGpu", [], {}),
                              ("ShortestPathBFSGpu", [rel(starts), rel(ends)], {}),
                              ("ClusteringCoefficientsGpu", [], {}), ("DegreeCentralityGpu", [], {})]:
        a = registry.run(name, [plain] + extra, opts)
        b = registry.run(name, [stored] + extra, opts)
        assert a == b and len(a) > 0, name


@pytest.mark.parametrize("n_key_cols", [3, 2])
def test_weighted_rules_read_stored_bytes(registry, n_key_cols):
    tuples = _edges(2, 40, 160, strings=True, weights=True)
    plain, stored = both(tuples, n_key_cols)
    nodes = sorted({t[0] for t in plain.iter()})
    starts, ends = [[nodes[0]], [nodes[5]]], [[nodes[2]], [nodes[9]]]
    for name, extra, opts in [("ShortestPathDijkstraGpu", [rel(starts)], {}),
                              ("ShortestPathDijkstraGpu", [rel(starts), rel(ends)], {"undirected": True}),
                              ("ClosenessCentralityGpu", [], {})]:
        a = registry.run(name, [plain] + extra, opts)
        b = registry.run(name, [stored] + extra, opts)
        assert a == b and len(a) > 0, name


def test_errors_and_fallbacks_match(registry):
    plain, stored = both([(1, 2, "heavy"), (2, 3, 1.0)], 2)
    for r in (plain, stored):
        with pytest.raises(FR.BadEdgeWeightError):
            registry.run("ShortestPathDijkstraGpu", [r, rel([[1]])], {})
    one_col = StoredInputRelation(codec.StoredRows.from_tuples(3, [(1,), (2,)], 1))
    with pytest.raises(FR.NotAnEdgeError):
        registry.run("PageRankGpu", [one_col], {})
    # a start node that has no edge: the ordered-id fast path hands over to the generic one, rows still equal
    tuples = _edges(4, 20, 50)
    plain, stored = both(tuples, 2)
    a = registry.run("ShortestPathBFSGpu", [plain, rel([[10 ** 6], [tuples[0][0]]]), rel([[tuples[5][1]]])], {})
    b = registry.run("ShortestPathBFSGpu", [stored, rel([[10 ** 6], [tuples[0][0]]]), rel([[tuples[5][1]]])], {})
    assert a == b and len(a) == 2
    assert stored.arity() == 2 and list(stored.iter()) == list(plain.iter())

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
                              ("ClosenessCentralityGpu", [], {}), ("BetweennessCentralityGpu", [], {"undirected": True})]:
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


# ---- HnswSearchRA over an index and a base relation that are read off their stored bytes -------------------------
class _OracleIndex:
    def __init__(self, O, flat):
        self.O, self.flat = O, flat

    def hnsw_knn_batch(self, queries, cfg):
        kk = cfg.ef if cfg.has_filter else cfg.k
        ids, dist, cnt, _ = self.flat.knn_batch(queries, kk, cfg.ef, radius=cfg.radius, dot_mode=self.O.DOT_GPU)
        return ids, dist, cnt


@pytest.mark.parametrize("backend", BACKENDS)
def test_hnsw_search_ra_over_stored_bytes(request, oracle, backend):
    from cozo_amd import build as B
    from cozo_amd.hnsw import BaseRelation, HnswIndexManifest, HnswSearchBinding, HnswSearchRA, index_nodes
    from cozo_amd.ingest import StoredHnswIndex, index_relation_tuples
    from cozo_amd.stored_relation import stored_base_relation
    B.build_ingest()
    rng = np.random.default_rng(3)
    dim = 16
    rows = [(f"doc-{i:04d}", i % 5, rng.random(dim, dtype=np.float32)) for i in range(300)]
    base = BaseRelation(keys=["id"], non_keys=["tag", "v"], rows=rows)
    nodes, vecs = index_nodes(base, [2])
    builder, flat = util.build_index(oracle, vecs, oracle.L2, 8, 40)
    tuples = index_relation_tuples([(rows[r][0], f, s) for r, f, s in nodes], vecs, flat.level_nodes, flat.level_nbrs, flat.entry,
                                   lambda p: oracle.distance_pairs(oracle.L2, vecs, vecs, p))
    s_idx, s_base = codec.StoredRows.from_tuples(8, tuples, 7), codec.StoredRows.from_tuples(7, rows, 1)
    got = StoredHnswIndex(s_idx, s_base, [2], dim, oracle.L2, 8)
    if backend == "oracle":
        ix_mem = _OracleIndex(oracle, flat)
        ix_sto = _OracleIndex(oracle, oracle.FlatIndex(got.vectors, oracle.L2, got.level_nodes, got.level_nbrs, got.entry))
    else:
        request.getfixturevalue("gpu_lib")
        ix_mem = util.gpu_index(flat, "L2", 8)
        ix_sto = got.to_gpu(HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=8))
    sb = HnswSearchBinding(k=4, ef=24, bind_field=True, bind_field_idx=True, bind_distance=True, bind_vector=True,
                           filter=lambda t: t[1] != 3)
    parent = [(i, rng.random(dim, dtype=np.float32)) for i in range(9)]
    a = HnswSearchRA(ix_mem, base, nodes, sb, bind_idx=1).iter(parent)
    b = HnswSearchRA(ix_sto, stored_base_relation(s_base, ["id"], ["tag", "v"]), got.nodes, sb, bind_idx=1).iter(parent)
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            assert np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v

"""HnswSearchRA::iter (query/ra.rs:1085-1121) + hnsw_knn's row assembly (runtime/hnsw.rs:939-1006) in the Python host
mirror: bind columns, radius, filter width, multi-vector rows.  The host logic runs on CPU with an oracle-backed
stand-in for the index handle (TEST ONLY), and through the GPU index on the device (marked gpu)."""
import numpy as np
import pytest

from cozo_amd.hnsw import BaseRelation, HnswSearch, HnswSearchBinding, HnswSearchRA, index_nodes
from tests import util


class OracleIndex:
    """hnsw_knn_batch of a GpuHnswIndex, computed by the oracle's flat index (kernel summation order)."""

    def __init__(self, O, flat):
        self.O, self.flat = O, flat

    def hnsw_knn_batch(self, queries, cfg: HnswSearch):
        kk = cfg.ef if cfg.has_filter else cfg.k
        ids, dist, cnt, _ = self.flat.knn_batch(queries, kk, cfg.ef, radius=cfg.radius, dot_mode=self.O.DOT_GPU)
        return ids, dist, cnt


BACKENDS = [pytest.param("oracle", id="host-logic"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


@pytest.fixture(scope="module")
def table(oracle):
    rng = np.random.default_rng(5)
    dim, n_rows = 16, 400
    rows = []
    for i in range(n_rows):
        v = rng.random(dim, dtype=np.float32)
        extra = [rng.random(dim, dtype=np.float32) for _ in range(i % 3)]  # a list column with 0..2 more vectors
        rows.append((i, f"row-{i}", v, extra))
    base = BaseRelation(keys=["k"], non_keys=["name", "v", "vs"], rows=rows)
    nodes, vecs = index_nodes(base, [2, 3])
    builder, flat = util.build_index(oracle, vecs, oracle.L2, 8, 40)
    return dict(base=base, nodes=nodes, vecs=vecs, flat=flat, dim=dim)


@pytest.fixture(params=BACKENDS)
def index(request, table, oracle):
    if request.param == "oracle":
        yield OracleIndex(oracle, table["flat"])
    else:
        request.getfixturevalue("gpu_lib")
        gix = util.gpu_index(table["flat"], "L2", 8)
        yield gix
        gix.close()


def literal_hnsw_knn(oracle, table, q, sb):
    """runtime/hnsw.rs:939-1006, one query, on the oracle's (node, distance) rows"""
    kk = sb.ef if sb.filter is not None else min(sb.k, sb.ef)
    ids, dist, cnt, _ = table["flat"].knn_batch(q[None, :], kk, sb.ef, dot_mode=oracle.DOT_GPU)
    out = []
    for j in range(int(cnt[0])):
        d = float(dist[0, j])
        if sb.radius is not None and d > sb.radius:
            continue
        r, f, s = table["nodes"][int(ids[0, j])]
        cand = list(table["base"].rows[r])
        if sb.bind_field:
            cand.append(table["base"].column_name(f))
        if sb.bind_field_idx:
            cand.append(None if s < 0 else s)
        if sb.bind_distance:
            cand.append(d)
        if sb.bind_vector:
            cand.append(cand[f] if s < 0 else cand[f][s])
        if sb.filter is not None and not sb.filter(tuple(cand)):
            continue
        out.append(tuple(cand))
    return out[:sb.k]


def same_rows(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            if isinstance(u, np.ndarray) or isinstance(v, np.ndarray):
                assert np.array_equal(u, v)
            elif isinstance(u, list):
                assert len(u) == len(v) and all(np.array_equal(p, q) for p, q in zip(u, v))
            else:
                assert u == v


@pytest.mark.parametrize("sb", [
    HnswSearchBinding(k=5, ef=20, bind_distance=True),
    HnswSearchBinding(k=3, ef=30, bind_field=True, bind_field_idx=True, bind_distance=True, bind_vector=True),
    HnswSearchBinding(k=4, ef=25, bind_distance=True, radius=0.9),
    HnswSearchBinding(k=4, ef=25, bind_distance=True, filter=lambda t: t[0] % 2 == 0),
    HnswSearchBinding(k=50, ef=10),  # k > ef: at most ef rows (found_nn holds ef entries)
], ids=["dist", "all-binds", "radius", "filter", "k>ef"])
def test_iter_matches_one_call_per_parent_tuple(index, table, oracle, sb):
    rng = np.random.default_rng(9)
    parent = [(f"q{i}", rng.random(table["dim"], dtype=np.float32)) for i in range(12)]
    ra = HnswSearchRA(index, table["base"], table["nodes"], sb, bind_idx=1)
    got = ra.iter(parent)
    want = []
    for t in parent:
        for c in literal_hnsw_knn(oracle, table, t[1], sb):
            want.append(tuple(t) + c)
    same_rows(got, want)
    assert got, "the case must produce rows"


def test_iter_rejects_non_vectors_and_handles_empty_parent(index, table):
    ra = HnswSearchRA(index, table["base"], table["nodes"], HnswSearchBinding(k=1, ef=1), bind_idx=0)
    assert ra.iter([]) == []
    with pytest.raises(ValueError):
        ra.iter([("not a vector",)])


def test_index_nodes_follow_hnsw_put_order(table):
    nodes = table["nodes"]
    assert nodes[0] == (0, 2, -1) and nodes[1] == (1, 2, -1) and nodes[2] == (1, 3, 0) and nodes[3] == (2, 2, -1)
    assert nodes[4] == (2, 3, 0) and nodes[5] == (2, 3, 1)
    assert len(nodes) == table["vecs"].shape[0] == 400 + sum(i % 3 for i in range(400))

"""libcozo_ingest.so (include/cozo_ingest.h): stored rows -> CSR / flat HNSW arrays, checked against the oracle's
restatement of as_directed_graph / CsrLayout::Sorted and the oracle-built index, and against the Python host mirror
(cozo_amd/fixed_rule.py, cozo_amd/codec.py) -- two independent restatements of the same reference formats."""
import numpy as np
import pytest

from cozo_amd import codec, ingest
from cozo_amd.fixed_rule import FixedRuleInputRelation
from cozo_amd.ingest import CozoIngestError, StoredGraph, StoredHnswIndex, index_relation_tuples
from tests import util

NONE = 0xFFFFFFFF


@pytest.fixture(scope="module", autouse=True)
def built():
    from cozo_amd import build as B
    B.build_ingest()


@pytest.fixture(params=["one-thread", "threaded"], autouse=True)
def id_assignment_path(request, monkeypatch):
    """every case runs through both forms of the first-appearance id assignment: the one-thread loop and the
    hash-partitioned one (forced on for small inputs here); they must produce the same ids bit for bit"""
    if request.param == "threaded":
        monkeypatch.setenv("CZI_THREADS", "5")
        monkeypatch.setenv("CZI_THREADED_MIN_ROWS", "0")
    else:
        monkeypatch.setenv("CZI_THREADS", "1")


@pytest.mark.parametrize("undirected", [False, True])
@pytest.mark.parametrize("n,e", [(50, 400), (2000, 9000), (3, 2)])
def test_int_relation_matches_oracle(oracle, n, e, undirected):
    frm, to = util.random_relation(n, e, n + e, self_loops=True)
    rows = codec.StoredRows.from_tuples(5, list(zip(frm.tolist(), to.tolist())), 2)
    g = StoredGraph(rows, undirected=undirected)
    fi, ti, ind = oracle.assign_ids(frm, to)
    assert g.n == len(ind) and g.indices() == ind.tolist()
    for inverse in (False, True):
        a, b = (ti, fi) if inverse else (fi, ti)
        want_off, want_tgt = oracle.build_csr(g.n, a, b, undirected=undirected)
        off, tgt, w = g.csr(inverse)
        assert w is None and np.array_equal(off, want_off) and np.array_equal(tgt, want_tgt)
    assert g.e == (2 if undirected else 1) * frm.size
    assert g.get_node_idx(int(ind[0])) == 0 and g.get_node_idx(float(ind[0])) is None and g.get_node_idx(10 ** 9) is None


def _value_rows(seed, n_rows):
    rng = np.random.default_rng(seed)
    pool = [None, True, False, 0, 1, 1.0, -3, 2.5, -0.0, 2 ** 60, "a", "b", "node-17", "", b"\x00\x01", b"", [1, "x"], [], [[2]],
            "a much longer string key than one group", 7, 7.0, float("inf")]
    rows = []
    for _ in range(n_rows):
        a, b = pool[rng.integers(len(pool))], pool[rng.integers(len(pool))]
        rows.append((a, b, float(rng.integers(0, 50)) / 4 if rng.random() < 0.5 else int(rng.integers(0, 50))))
    return rows


@pytest.mark.parametrize("n_key_cols", [3, 2, 1, 0])
@pytest.mark.parametrize("undirected", [False, True])
def test_mixed_values_match_python_host_mirror(n_key_cols, undirected):
    """endpoints of every DataValue kind, living in the key part or in the msgpack value part of the row: ids, CSR and
    weights equal those of FixedRuleInputRelation over the decoded tuples"""
    tuples = _value_rows(3, 300)
    rows = codec.StoredRows.from_tuples(9, tuples, n_key_cols)
    decoded = rows.tuples()  # scan order; later puts of a key replaced earlier ones
    want_g, want_ind, _ = FixedRuleInputRelation(decoded, arity=3).as_directed_weighted_graph(undirected, False)
    g = StoredGraph(rows, undirected=undirected, weighted=True)
    got_ind = g.indices()
    assert len(got_ind) == len(want_ind) == g.n
    for x, y in zip(got_ind, want_ind):
        assert type(x) is type(y) and (x == y) and str(x) == str(y)
    off, tgt, w = g.csr(False)
    assert np.array_equal(off, want_g.out_offsets) and np.array_equal(tgt, want_g.out_targets)
    assert np.array_equal(w, want_g.out_weights)
    off, tgt, _ = g.csr(True)
    assert np.array_equal(off, want_g.in_offsets) and np.array_equal(tgt, want_g.in_sources)
    for i, v in enumerate(want_ind):
        assert g.get_node_idx(v) == i


def test_rows_in_the_scan_are_ordered_by_key_bytes_like_the_relation():
    """FixedRuleInputRelation sorts its rows by DataValue order; a stored relation by key bytes: same scan"""
    tuples = [(a, b) for a, b, _ in _value_rows(5, 200)]
    rows = codec.StoredRows.from_tuples(1, tuples, 2)
    rel = FixedRuleInputRelation(tuples, arity=2)
    want_g, want_ind, _ = rel.as_directed_graph(False)
    g = StoredGraph(rows)
    assert [str(x) for x in g.indices()] == [str(x) for x in want_ind]
    off, tgt, _ = g.csr(False)
    assert np.array_equal(off, want_g.out_offsets) and np.array_equal(tgt, want_g.out_targets)


def test_graph_errors():
    with pytest.raises(CozoIngestError) as e:
        StoredGraph(codec.StoredRows.from_tuples(1, [(1,), (2,)], 1))
    assert e.value.code == ingest.CZI_E_NOT_AN_EDGE  # NotAnEdgeError
    for bad in ("heavy", None, float("inf"), float("nan"), -1.0, -1):
        with pytest.raises(CozoIngestError) as e:
            StoredGraph(codec.StoredRows.from_tuples(1, [(1, 2, 1.0), (2, 3, bad)], 2), weighted=True)
        assert e.value.code == ingest.CZI_E_BAD_WEIGHT
        with pytest.raises(CozoIngestError) as e:
            StoredGraph(codec.StoredRows.from_tuples(1, [(1, 2, 1.0), (2, 3, bad)], 3), weighted=True)
        assert e.value.code == ingest.CZI_E_BAD_WEIGHT
    g = StoredGraph(codec.StoredRows.from_tuples(1, [(1, 2, -1.5)], 2), weighted=True, allow_negative_weights=True)
    assert g.csr()[2].tolist() == [-1.5]
    g = StoredGraph(codec.StoredRows.from_tuples(1, [(1, 2), (2, 3)], 2), weighted=True)  # no third column: 1.0
    assert g.csr()[2].tolist() == [1.0, 1.0]
    assert StoredGraph(codec.StoredRows.from_tuples(1, [], 2)).n == 0
    rows = codec.StoredRows.from_tuples(1, [(1, 2)], 2)
    rows.keys = rows.keys[:-3]
    rows.key_off = rows.key_off.copy()
    rows.key_off[-1] -= 3
    with pytest.raises(CozoIngestError) as e:
        StoredGraph(rows)
    assert e.value.code == ingest.CZI_E_CORRUPT


def test_variant_index_form_of_values_is_accepted():
    """a serde encoder that writes variant indices instead of names ({2: {0: 5}}) decodes to the same graph"""
    import msgpack
    import struct
    names = codec.StoredRows.from_tuples(4, [(1, 2, 3.5), (2, "x", 1)], 1)
    alt_vals = [struct.pack(">Q", 4) + msgpack.packb([{2: {0: 2}}, {2: {1: 3.5}}]),
                struct.pack(">Q", 4) + msgpack.packb([{3: "x"}, {2: {0: 1}}])]
    alt = codec.StoredRows(names.keys, names.key_off, b"".join(alt_vals),
                           np.array([0, len(alt_vals[0]), len(alt_vals[0]) + len(alt_vals[1])], dtype=np.uint64), 1)
    a, b = StoredGraph(names, weighted=True), StoredGraph(alt, weighted=True)
    assert a.indices() == b.indices() == [1, 2, "x"]
    for x, y in zip(a.csr(), b.csr()):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------------------------------------------- HNSW
def _index_case(oracle, multi):
    rng = np.random.default_rng(11)
    dim, n_rows = 8, 260
    rows = []
    for i in range(n_rows):
        v = rng.random(dim, dtype=np.float32)
        extra = [rng.random(dim, dtype=np.float32) for _ in range(i % 3)] if multi else []
        rows.append((i * 3 - 100, f"row-{i}", v, extra))  # negative and positive int keys: byte order == numeric order
    from cozo_amd.hnsw import BaseRelation, index_nodes
    base = BaseRelation(keys=["k"], non_keys=["name", "v", "vs"], rows=rows)
    nodes, vecs = index_nodes(base, [2, 3])
    builder, flat = util.build_index(oracle, vecs, oracle.L2, 6, 30)
    key_of_node = [(rows[r][0], f, s) for r, f, s in nodes]

    def link_distance(pairs):
        return oracle.distance_pairs(oracle.L2, vecs, vecs, pairs)
    tuples = index_relation_tuples(key_of_node, vecs, flat.level_nodes, flat.level_nbrs, flat.entry, link_distance, relation_id=21)
    return dict(rows=rows, nodes=nodes, vecs=vecs, flat=flat, tuples=tuples, dim=dim,
                idx=codec.StoredRows.from_tuples(21, tuples, 7), base=codec.StoredRows.from_tuples(20, rows, 1))


def test_index_relation_round_trip(oracle):
    """flat index -> `tbl:idx` tuples -> stored bytes -> flat index: identical tables, entry point and vectors"""
    c = _index_case(oracle, multi=False)
    flat = c["flat"]
    got = StoredHnswIndex(c["idx"], c["base"], [2, 3], c["dim"], oracle.L2, 6)
    assert (got.n, got.dim, got.n_levels, got.entry) == (flat.n, flat.dim, flat.n_levels, flat.entry)
    assert np.array_equal(got.vectors, flat.vectors)
    assert got.nodes == [(r, f, s) for r, f, s in c["nodes"]]
    for lv in range(flat.n_levels):
        assert got.level_width[lv] == flat.level_width[lv]
        assert np.array_equal(got.level_nodes[lv], flat.level_nodes[lv])
        assert np.array_equal(got.level_nbrs[lv], flat.level_nbrs[lv])
    # structural counters, as runtime/tests.rs:730,737 count rows of the index relation
    live = sum(int((t != NONE).sum()) for t in flat.level_nbrs)
    assert (got.n_self, got.n_live_links, got.n_ignored) == (sum(int(s) for s in flat.level_size), live, 0)
    assert got.n_rows == got.n_self + live + 1  # + the canary row
    # the reference's own reading of the tuples: the first row in key order names the entry point on the top layer
    first = c["idx"].tuples()[0]
    assert first[0] == -(flat.n_levels - 1) and tuple(first[1:4]) == (c["rows"][c["nodes"][flat.entry][0]][0],) + tuple(c["nodes"][flat.entry][1:])


def test_index_rows_dropped_like_hnsw_get_neighbours(oracle):
    """links between two vectors of one base row and ignore_link rows are not neighbours (hnsw.rs:609-624)"""
    c = _index_case(oracle, multi=True)
    flat, nodes = c["flat"], c["nodes"]
    tuples = [list(t) for t in c["tuples"]]
    flagged = set()
    k = 0
    for t in tuples:  # soft-delete every 7th link row
        if t[0] <= 0 and tuple(t[1:4]) != tuple(t[4:7]):
            k += 1
            if k % 7 == 0:
                t[9] = True
                flagged.add((t[0], tuple(t[1:4]), tuple(t[4:7])))
    idx = codec.StoredRows.from_tuples(21, tuples, 7)
    got = StoredHnswIndex(idx, c["base"], [2, 3], c["dim"], oracle.L2, 6)
    key_of = [(c["rows"][r][0], f, s) for r, f, s in nodes]
    n_same_row = 0
    for lv in range(flat.n_levels):
        for r, fr in enumerate(flat.level_nodes[lv]):
            want = []
            for t in flat.level_nbrs[lv][r]:
                if t == NONE:
                    continue
                if nodes[t][0] == nodes[fr][0]:
                    n_same_row += 1
                    continue
                if (-lv, key_of[fr], key_of[t]) in flagged:
                    continue
                want.append(int(t))
            row = got.level_nbrs[lv][r]
            assert row[row != NONE].tolist() == want
    assert got.n_ignored > 0 and n_same_row > 0  # the case exercises both rules
    assert np.array_equal(got.vectors, c["vecs"])  # list elements and plain vectors, all from the msgpack value part


def test_vectors_in_key_columns_and_empty_index(oracle):
    rng = np.random.default_rng(2)
    dim = 4
    vecs = rng.random((40, dim), dtype=np.float32)
    rows = [(i, vecs[i]) for i in range(40)]  # both columns are keys: the vector sits in the key (big-endian there)
    builder, flat = util.build_index(oracle, vecs, oracle.L2, 4, 20)
    tuples = index_relation_tuples([(i, vecs[i], 1, -1) for i in range(40)], vecs, flat.level_nodes, flat.level_nbrs, flat.entry,
                                   lambda p: oracle.distance_pairs(oracle.L2, vecs, vecs, p))
    idx = codec.StoredRows.from_tuples(2, tuples, 2 * 2 + 5)
    base = codec.StoredRows.from_tuples(1, rows, 2)
    # ids follow the KEY order of the rows = the order of the ints here
    got = StoredHnswIndex(idx, base, [1], dim, oracle.L2, 4)
    assert np.array_equal(got.vectors, vecs) and got.entry == flat.entry
    for lv in range(flat.n_levels):
        assert np.array_equal(got.level_nbrs[lv], flat.level_nbrs[lv])
    # an index holding only the canary row, or nothing at all: empty (hnsw.rs:903-909)
    canary_only = codec.StoredRows.from_tuples(2, [tuples[-1]], 9)
    assert tuples[-1][0] == 1
    for stored in (canary_only, codec.StoredRows.from_tuples(2, [], 9)):
        e = StoredHnswIndex(stored, base, [1], dim, oracle.L2, 4)
        assert (e.n, e.n_levels) == (0, 0)


def test_hnsw_errors(oracle):
    c = _index_case(oracle, multi=False)
    with pytest.raises(CozoIngestError) as e:  # a base row is gone: "corrupted index"
        StoredHnswIndex(c["idx"], codec.StoredRows.from_tuples(20, c["rows"][1:], 1), [2, 3], c["dim"], oracle.L2, 6)
    assert e.value.code == ingest.CZI_E_MISSING_ROW
    with pytest.raises(CozoIngestError) as e:  # wrong dimension
        StoredHnswIndex(c["idx"], c["base"], [2, 3], c["dim"] + 1, oracle.L2, 6)
    assert e.value.code == ingest.CZI_E_CORRUPT
    with pytest.raises(CozoIngestError) as e:  # key column count of the index relation does not fit the base relation
        StoredHnswIndex(codec.StoredRows(c["idx"].keys, c["idx"].key_off, c["idx"].vals, c["idx"].val_off, 9), c["base"], [2, 3],
                        c["dim"], oracle.L2, 6)
    assert e.value.code == ingest.CZI_E_INVALID
    f64rows = [(r[0], r[1], r[2].astype(np.float64), r[3]) for r in c["rows"]]
    with pytest.raises(CozoIngestError) as e:
        StoredHnswIndex(c["idx"], codec.StoredRows.from_tuples(20, f64rows, 1), [2, 3], c["dim"], oracle.L2, 6)
    assert e.value.code == ingest.CZI_E_UNSUPPORTED


def test_f64_index_keeps_its_vectors_as_f64(oracle):
    """an index whose manifest says VecElementType::F64: czi_hnsw_ingest_f64 hands the stored f64 vectors over
    bit for bit (values NOT representable in f32), from value columns (msgpack) and from a key column (memcmp bytes) alike; the
    link tables are what the F32 form yields for the same rows; an F32 row in an F64 index (or the reverse) is refused"""
    c = _index_case(oracle, multi=True)
    rng = np.random.default_rng(5)

    def wide(v):
        return v.astype(np.float64) + rng.standard_normal(v.shape) * 1e-9

    rows64 = [(r[0], r[1], wide(r[2]), [wide(x) for x in r[3]]) for r in c["rows"]]
    base64 = codec.StoredRows.from_tuples(20, rows64, 1)
    got = StoredHnswIndex(c["idx"], base64, [2, 3], c["dim"], oracle.L2, 6, dtype="F64")
    ref = StoredHnswIndex(c["idx"], c["base"], [2, 3], c["dim"], oracle.L2, 6)
    assert got.dtype == "F64" and got.vectors.dtype == np.float64 and got.nodes == ref.nodes and got.entry == ref.entry
    want = np.stack([rows64[r][f] if s < 0 else rows64[r][f][s] for r, f, s in got.nodes])
    assert np.array_equal(got.vectors, want) and not np.array_equal(want, want.astype(np.float32).astype(np.float64))
    for lv in range(ref.n_levels):
        assert np.array_equal(got.level_nodes[lv], ref.level_nodes[lv]) and np.array_equal(got.level_nbrs[lv], ref.level_nbrs[lv])
    with pytest.raises(CozoIngestError) as e:  # f32 rows under an F64 manifest
        StoredHnswIndex(c["idx"], c["base"], [2, 3], c["dim"], oracle.L2, 6, dtype="F64")
    assert e.value.code == ingest.CZI_E_UNSUPPORTED
    # the vector as the (only) key column: its memcmp form, big-endian raw f64 elements (data/memcmp.rs:62-68)
    dim, n = 3, 40
    vecs = rng.standard_normal((n, dim))
    vecs = vecs[np.lexsort(vecs.T[::-1])]
    krows = [(vecs[i], i) for i in range(n)]
    from cozo_amd.hnsw import BaseRelation, index_nodes
    kb = BaseRelation(keys=["v"], non_keys=["i"], rows=krows)
    nodes, v32 = index_nodes(kb, [0])
    builder, flat = util.build_index(oracle, np.asarray(v32, dtype=np.float32), oracle.L2, 4, 20)
    key_of_node = [(krows[r][0], f, s2) for r, f, s2 in nodes]
    tuples = index_relation_tuples(key_of_node, np.asarray(v32, dtype=np.float32), flat.level_nodes, flat.level_nbrs, flat.entry,
                                   lambda pairs: oracle.distance_pairs(oracle.L2, np.asarray(v32, np.float32), np.asarray(v32, np.float32), pairs),
                                   relation_id=31)
    gk = StoredHnswIndex(codec.StoredRows.from_tuples(31, tuples, 7), codec.StoredRows.from_tuples(30, krows, 1), [0], dim, oracle.L2, 4,
                         dtype="F64")
    # (node ids follow the stored key order of the index rows, base rows theirs: compare as sets of bit patterns)
    assert gk.vectors.dtype == np.float64 and gk.n == n
    assert sorted(v.tobytes() for v in gk.vectors) == sorted(v.tobytes() for v in vecs)


@pytest.mark.parametrize("n_key_cols", [2, 1])
def test_ordered_ids_are_value_ranks(n_key_cols):
    """CZI_ORDERED_IDS == FixedRuleInputRelation.as_ordered_graph: ids by DataValue order, so that ascending id in a CSR
    row is the key order ShortestPathBFS / Bfs meet neighbours in"""
    tuples = [(a, b) for a, b, _ in _value_rows(8, 250)]
    rows = codec.StoredRows.from_tuples(1, tuples, n_key_cols)
    want_g, want_ind, _ = FixedRuleInputRelation(rows.tuples(), arity=2).as_ordered_graph()
    g = StoredGraph(rows, ordered_ids=True)
    assert [str(x) for x in g.indices()] == [str(x) for x in want_ind]
    off, tgt, _ = g.csr(False)
    assert np.array_equal(off, want_g.out_offsets) and np.array_equal(tgt, want_g.out_targets)
    off, tgt, _ = g.csr(True)
    assert np.array_equal(off, want_g.in_offsets) and np.array_equal(tgt, want_g.in_sources)
    for i, v in enumerate(want_ind):
        assert g.get_node_idx(v) == i


@pytest.mark.parametrize("multi", [False, True])
def test_native_row_encoder_equals_python_codec(oracle, multi):
    """czi_hnsw_encode_rows (hand-written msgpack / memcmp / SHA-256) == index_relation_tuples + cozo_amd/codec.py
    (the msgpack package, hashlib), byte for byte, keys and values, in key order"""
    from cozo_amd.ingest import encode_index_rows
    c = _index_case(oracle, multi)
    flat, vecs, nodes = c["flat"], c["vecs"], c["nodes"]
    key_of_node = [(c["rows"][r][0], f, s) for r, f, s in nodes]
    level_dist = []
    for lv in range(flat.n_levels):
        tab = flat.level_nbrs[lv]
        ids = flat.level_nodes[lv]
        dist = np.zeros(tab.shape, dtype=np.float64)
        for r in range(tab.shape[0]):
            live = np.nonzero(tab[r] != NONE)[0]
            pairs = np.stack([np.full(live.size, ids[r], dtype=np.uint32), tab[r][live]], 1)
            dist[r, live] = oracle.distance_pairs(oracle.L2, vecs, vecs, pairs)
        level_dist.append(dist)
    got = encode_index_rows(key_of_node, vecs, flat.level_nodes, flat.level_nbrs, flat.entry, oracle.L2, level_dist, 21)
    want = c["idx"]
    assert len(got) == len(want) and got.n_key_cols == want.n_key_cols == 7
    assert np.array_equal(got.key_off, want.key_off) and got.keys == want.keys
    assert np.array_equal(got.val_off, want.val_off) and got.vals == want.vals
    # and it reads back as the index it came from
    back = StoredHnswIndex(got, c["base"], [2, 3], c["dim"], oracle.L2, 6)
    assert back.entry == flat.entry and back.n == flat.n


def test_sha256_of_the_row_encoder_known_answers():
    """FIPS 180-4 vectors through the self-loop row's hash column: a vector whose little-endian bytes are the message"""
    import hashlib
    from cozo_amd.ingest import encode_index_rows
    for dim in (1, 13, 14, 16, 31, 200):  # 4*dim bytes: below / at / above the 55-byte and 64-byte padding boundaries
        vecs = np.arange(2 * dim, dtype=np.float32).reshape(2, dim) / 7
        nb = [np.array([[1, NONE], [0, NONE]], dtype=np.uint32)]
        rows = encode_index_rows([(0, 1, -1), (1, 1, -1)], vecs, [None], nb, 0, 0, [np.zeros((2, 2))], 5)
        tup = rows.tuples()
        selfs = [t for t in tup if t[0] == 0 and t[1:4] == t[4:7]]
        assert [t[8] for t in selfs] == [hashlib.sha256(vecs[i].tobytes()).digest() for i in range(2)]
        assert [t[7] for t in selfs] == [1.0, 1.0]


def test_row_encoder_takes_the_degrees_of_the_self_rows():
    """czi_hnsw_encode_rows_degrees: the f64 of a self row is what the caller hands in (cz_hnsw_index_export_degrees: with
    extend_candidates a shrink leaves it one above the number of link rows, hnsw.rs:413-433 + 352-357); without, the link count"""
    from cozo_amd.ingest import encode_index_rows
    vecs = np.arange(12, dtype=np.float32).reshape(3, 4)
    nb = [np.array([[1, 2], [0, NONE], [0, NONE]], dtype=np.uint32), np.array([[NONE]], dtype=np.uint32)]
    nodes = [None, np.array([1], dtype=np.uint32)]
    keys = [(10, 1, -1), (11, 1, -1), (12, 1, -1)]
    dist = [np.ones((3, 2)), np.zeros((1, 1))]
    plain = encode_index_rows(keys, vecs, nodes, nb, 1, 0, dist, 5).tuples()
    given = encode_index_rows(keys, vecs, nodes, nb, 1, 0, dist, 5, level_degree=[np.array([3.0, 1.0, 2.0]), np.array([0.0])]).tuples()
    selfs = lambda tup: {(t[0], t[1]): t[7] for t in tup if t[0] <= 0 and t[1:4] == t[4:7]}
    assert selfs(plain) == {(0, 10): 2.0, (0, 11): 1.0, (0, 12): 1.0, (-1, 11): 0.0}
    assert selfs(given) == {(0, 10): 3.0, (0, 11): 1.0, (0, 12): 2.0, (-1, 11): 0.0}
    strip = lambda tup: [t for t in tup if not (t[0] <= 0 and t[1:4] == t[4:7])]
    assert strip(plain) == strip(given)


def _dump(rows: codec.StoredRows, path):
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<IQ", rows.n_key_cols, len(rows)))
        f.write(np.ascontiguousarray(rows.key_off, dtype="<u8").tobytes())
        f.write(np.ascontiguousarray(rows.val_off, dtype="<u8").tobytes())
        f.write(struct.pack("<Q", len(rows.keys)) + rows.keys)
        f.write(struct.pack("<Q", len(rows.vals)) + rows.vals)


def test_parsers_survive_damaged_rows_under_sanitizers(oracle, tmp_path):
    """tests/cpp/fuzz_ingest.cpp: the library's sources built with -fsanitize=address,undefined and with -fsanitize=thread, thousands of
    mutated inputs (byte flips, shifted row boundaries, wrong column counts): a status code every time, never a fault or race"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = _index_case(oracle, multi=True)
    graph = codec.StoredRows.from_tuples(9, _value_rows(3, 200), 2)
    for name, rows in (("graph", graph), ("idx", c["idx"]), ("base", c["base"])):
        _dump(rows, tmp_path / f"{name}.bin")
    from cozo_amd.build import INGEST_DIR, ingest_sources
    src = [os.path.join(root, "tests", "cpp", "fuzz_ingest.cpp"), *ingest_sources()]
    deps = src + [os.path.join(INGEST_DIR, "common.hpp")]
    os.makedirs(os.path.join(root, "tests", "cpp", "bin"), exist_ok=True)
    args = [str(tmp_path / "graph.bin"), str(tmp_path / "idx.bin"), str(tmp_path / "base.bin")]
    # address + undefined-behaviour sanitizers: the one-thread path and the hash-partitioned one; thread sanitizer: the latter
    for san, name, runs in (("address,undefined", "fuzz_ingest", (("1", "1500"), ("3", "700"))), ("thread", "fuzz_ingest_tsan", (("4", "300"),))):
        exe = os.path.join(root, "tests", "cpp", "bin", name)
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(s) for s in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + san, "-fno-sanitize-recover=all",
                                   "-I" + os.path.join(root, "include"), *src, "-o", exe])
        for threads, iters in runs:
            env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1", LD_PRELOAD="",
                       CZI_THREADS=threads, CZI_THREADED_MIN_ROWS="0")
            res = subprocess.run([exe, *args, iters], capture_output=True, text=True, env=env, timeout=900)
            assert res.returncode == 0, (san, threads, res.stdout[-2000:] + res.stderr[-4000:])
            assert "no fault" in res.stdout


def test_threaded_and_one_thread_agree_at_scale(monkeypatch):
    """0.6 M rows (past the default threshold): the hash-partitioned numbering and the position-split radix CSR against the
    one-thread forms -- node keys, ids, offsets, targets and weights identical"""
    rng = np.random.default_rng(17)
    e = 600_000
    pairs = np.unique(rng.integers(0, 70_000, (e, 2), dtype=np.int64), axis=0)
    n = pairs.shape[0]
    w = (rng.integers(0, 64, n) / 8).astype(np.float64)
    # (int, int) keys, f64 weight in the value part; built vectorised: 28-byte keys, 25-byte values
    rec = np.zeros((n, 28), dtype=np.uint8)
    rec[:, 7] = 3
    for c in range(2):
        img = pairs[:, c].astype(np.float64).view(np.uint64) | np.uint64(0x8000000000000000)
        rec[:, 8 + 10 * c] = 0x05
        rec[:, 9 + 10 * c:17 + 10 * c] = img.byteswap().view(np.uint8).reshape(n, 8)
    head = bytes([0, 0, 0, 0, 0, 0, 0, 3, 0x91, 0x81, 0xa3]) + b"Num" + bytes([0x81, 0xa5]) + b"Float" + bytes([0xcb])
    val = np.zeros((n, len(head) + 8), dtype=np.uint8)
    val[:, :len(head)] = np.frombuffer(head, dtype=np.uint8)
    val[:, len(head):] = w.view(np.uint64).byteswap().view(np.uint8).reshape(n, 8)
    rows = codec.StoredRows(rec.tobytes(), np.arange(n + 1, dtype=np.uint64) * 28, val.tobytes(),
                            np.arange(n + 1, dtype=np.uint64) * val.shape[1], 2)
    assert rows.tuples()[:1] == [[int(pairs[0, 0]), int(pairs[0, 1]), float(w[0])]]  # the fabricated bytes are real rows
    out = {}
    for threads in ("1", "6"):
        monkeypatch.setenv("CZI_THREADS", threads)
        monkeypatch.delenv("CZI_THREADED_MIN_ROWS", raising=False)
        g = StoredGraph(rows, undirected=True, weighted=True)
        out[threads] = (g.n, g.node_keys(), g.csr(False), g.csr(True), g.get_node_idx(int(pairs[n // 2, 1])))
        g.close()
    a, b = out["1"], out["6"]
    assert a[0] == b[0] and a[1][0] == b[1][0] and np.array_equal(a[1][1], b[1][1]) and a[4] == b[4]
    for x, y in zip(a[2] + a[3], b[2] + b[3]):
        assert (x is None and y is None) or np.array_equal(x, y)
    assert a[0] <= 70_000 and a[2][1].size == 2 * n


def test_reference_pinned_index_row_counts():
    """runtime/tests.rs:700-741 `test_vec_index_insertion`: of two rows only 'a' passes the index filter; the reference asserts
    that `?[k] := *a:vec{layer: 0, fr_k, to_k}, k = fr_k or k = to_k` then has exactly ONE row, and none once 'a' leaves the
    index.  The rows the write-back emits for that one-vector index answer the same query the same way."""
    vecs = np.array([[1, 2]], dtype=np.float32)
    nbrs = [np.full((1, 100), NONE, dtype=np.uint32)]  # m = 50: level 0 rows are 2m wide, no links yet
    tuples = index_relation_tuples([("a", 1, -1)], vecs, [None], nbrs, 0, lambda p: np.zeros(len(p)))
    layer0 = [t for t in tuples if t[0] == 0]
    assert {t[1] for t in layer0} | {t[4] for t in layer0} == {"a"}  # the query's distinct k: one row
    assert len(tuples) == 2 and tuples[-1][0] == 1  # the self-loop row + the canary
    assert layer0[0][7] == 0.0 and len(layer0[0][8]) == 32 and layer0[0][9] is False  # degree 0, SHA-256, not ignored
    idx = codec.StoredRows.from_tuples(2, tuples, 7)
    base = codec.StoredRows.from_tuples(1, [("a", vecs[0], True), ("b", np.array([2, 3], np.float32), False)], 1)
    got = StoredHnswIndex(idx, base, [1], 2, 0, 50)
    assert (got.n, got.n_levels, got.entry, got.nodes) == (1, 1, 0, [(0, 1, -1)]) and np.array_equal(got.vectors, vecs)
    assert got.level_nbrs[0].shape == (1, 100) and (got.level_nbrs[0] == NONE).all()
    # 'a' leaves the index: no rows at layer 0 (what is left of an emptied index is at most the canary)
    emptied = StoredHnswIndex(codec.StoredRows.from_tuples(2, [tuples[-1]], 7), base, [1], 2, 0, 50)
    assert emptied.n == 0 and emptied.n_levels == 0

"""Property tests (hypothesis) of the stored-row formats: arbitrary nested values round-trip through the memcmp key encoding
and the msgpack value encoding, byte order is value order, and libcozo_ingest (C++) numbers arbitrary relations exactly like
the host mirror numbers their decoded tuples."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from cozo_amd import codec
from cozo_amd.fixed_rule import FixedRuleInputRelation, sort_key

scalars = st.one_of(
    st.none(), st.booleans(), st.integers(min_value=-(2 ** 63), max_value=2 ** 63 - 1),
    st.floats(allow_nan=False), st.text(max_size=20), st.binary(max_size=20))
values = st.recursive(scalars, lambda inner: st.lists(inner, max_size=4), max_leaves=8)


def same(x, y):
    if isinstance(x, list):
        return isinstance(y, list) and len(x) == len(y) and all(same(a, b) for a, b in zip(x, y))
    return type(x) is type(y) and (x == y) and str(x) == str(y)


@settings(max_examples=300, deadline=None)
@given(values)
def test_memcmp_round_trip(v):
    b = codec.memcmp_bytes(v)
    dec, at = codec.decode_datavalue(b)
    assert at == len(b) and same(dec, v)


@settings(max_examples=300, deadline=None)
@given(values, values)
def test_byte_order_is_value_order(a, b):
    ka, kb = sort_key(a), sort_key(b)
    ea, eb = codec.memcmp_bytes(a), codec.memcmp_bytes(b)
    assert (ka < kb) == (ea < eb) and (ka == kb) == (ea == eb)


@settings(max_examples=200, deadline=None)
@given(st.lists(values, min_size=1, max_size=5), st.integers(min_value=0, max_value=5))
def test_row_round_trip_through_the_store(row, n_key_cols):
    k = codec.encode_key_for_store(5, row[:n_key_cols])
    v = codec.encode_val_for_store(5, row[n_key_cols:])
    back = codec.decode_tuple_from_kv(k, v)
    assert len(back) == len(row) and all(same(x, y) for x, y in zip(back, row))


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.lists(st.tuples(values, values, st.one_of(st.integers(0, 100), st.floats(0, 100))), min_size=1, max_size=40),
       st.integers(min_value=0, max_value=3), st.booleans(), st.sampled_from(["1", "3"]))
def test_ingest_numbers_relations_like_the_host_mirror(rows, n_key_cols, undirected, threads):
    import os
    from cozo_amd import build as B
    from cozo_amd.ingest import StoredGraph
    B.build_ingest()
    os.environ["CZI_THREADS"], os.environ["CZI_THREADED_MIN_ROWS"] = threads, "0"
    try:
        stored = codec.StoredRows.from_tuples(2, rows, n_key_cols)
        want_g, want_ind, _ = FixedRuleInputRelation([tuple(t) for t in stored.tuples()], arity=3).as_directed_weighted_graph(undirected, False)
        g = StoredGraph(stored, undirected=undirected, weighted=True)
        got_ind = g.indices()
        assert len(got_ind) == len(want_ind) and all(same(x, y) for x, y in zip(got_ind, want_ind))
        off, tgt, w = g.csr(False)
        assert np.array_equal(off, want_g.out_offsets) and np.array_equal(tgt, want_g.out_targets) and np.array_equal(w, want_g.out_weights)
        off, tgt, _ = g.csr(True)
        assert np.array_equal(off, want_g.in_offsets) and np.array_equal(tgt, want_g.in_sources)
    finally:
        os.environ.pop("CZI_THREADS", None)
        os.environ.pop("CZI_THREADED_MIN_ROWS", None)

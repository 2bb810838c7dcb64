"""cozo_amd/codec.py: the memcmp key encoding and the stored row forms.  The property tests restate the reference's
own (data/tests/memcmp.rs:15-137): round trips, and byte order == value order."""
import functools
import uuid

import numpy as np
import pytest

from cozo_amd import codec
from cozo_amd.fixed_rule import sort_key


def enc(v):
    return codec.memcmp_bytes(v)


def test_encode_decode_num_and_order():
    """data/tests/memcmp.rs:15-51: ints around every power of two down to 2^9, +-inf, NaN, random floats and their
    reciprocals; decode(encode(x)) == x and sorting the encodings sorts the numbers (Int before the equal Float)"""
    rng = np.random.default_rng(1)
    nums = []
    n = (1 << 63) - 1
    for i in range(54):
        for j in range(0, 1000, 37):
            vb = (n >> i) - j
            nums += [vb, -vb - 1]
    nums += [float("inf"), float("-inf")]
    for _ in range(3000):
        f = (rng.random() - 0.5) * 2.0
        nums += [f, 1.0 / f]
    nums += [0, 0.0, -0.0, 1, 1.0, -1, -1.0, 2 ** 53, float(2 ** 53), 2 ** 53 + 1, -(2 ** 53), -(2 ** 53) - 1]
    encoded = []
    for x in nums:
        out = bytearray()
        codec.encode_num(out, x)
        y, at = codec.decode_num(bytes(out), 0)
        assert at == len(out) and type(y) is type(x) and (y == x)
        assert str(y) == str(x)  # -0.0 stays -0.0
        encoded.append((bytes(out), x))
    by_bytes = [x for _, x in sorted(encoded, key=lambda t: t[0])]
    by_value = [x for _, x in sorted(encoded, key=lambda t: (sort_key(t[1]), str(t[1])))]
    assert [sort_key(x) for x in by_bytes] == [sort_key(x) for x in by_value]
    out = bytearray()
    codec.encode_num(out, float("nan"))
    assert np.isnan(codec.decode_num(bytes(out), 0)[0])
    assert bytes(out) > max(e for e, _ in encoded)  # NaN (positive) sorts after +inf, like f64::total_cmp


def test_exact_int_bound():
    """memcmp.rs:130-140: |i| < 2^53 is carried by the float part alone (10 bytes with the tag), beyond that 8 more"""
    assert len(enc(2 ** 53 - 1)) == 10 and len(enc(-(2 ** 53) + 1)) == 10
    assert len(enc(2 ** 53)) == 18 and len(enc(-(2 ** 53))) == 18
    assert enc(1) < enc(1.0) < enc(2)  # Num::cmp: Int < the equal Float (data/value.rs:578-593)
    for i in (2 ** 63 - 1, 2 ** 63 - 2, -(2 ** 63), -(2 ** 63) + 1):  # neighbours that round to the same f64
        assert codec.decode_datavalue(enc(i))[0] == i
    assert enc(2 ** 63 - 2) < enc(2 ** 63 - 1) and enc(-(2 ** 63)) < enc(-(2 ** 63) + 1)


def test_encode_decode_bytes():
    """data/tests/memcmp.rs:65-98"""
    target = b"Lorem ipsum dolor sit amet, consectetur adipiscing elit..."
    for i in range(len(target)):
        bs = target[i:]
        out = bytearray()
        codec.encode_bytes(out, bs)
        dec, at = codec.decode_bytes(bytes(out))
        assert dec == bs and at == len(out) and len(out) == (len(bs) // 8 + 1) * 9
        out = bytearray()
        for part in (target, bs, bs, target):
            codec.encode_bytes(out, part)
        at = 0
        for part in (target, bs, bs, target):
            dec, at = codec.decode_bytes(bytes(out), at)
            assert dec == part
        assert at == len(out)
    strings = [b"", b"a", b"a\x00", b"a\x00\x00", b"ab", b"abcdefgh", b"abcdefgh\x00", b"abcdefghi", b"b"]
    encs = []
    for s in strings:
        out = bytearray()
        codec.encode_bytes(out, s)
        encs.append(bytes(out))
    assert sorted(encs) == [encs[strings.index(s)] for s in sorted(strings)]  # prefix-free and order preserving


def test_encode_decode_uuid_and_specific():
    """data/tests/memcmp.rs:53-63, 100-113"""
    u = uuid.UUID("dd85b19a-5fde-11ed-a88e-1774a7698039")
    b = enc(u)
    assert len(b) == 17 and b[1:3] == bytes.fromhex("11ed") and b[3:5] == bytes.fromhex("5fde") and b[5:9] == bytes.fromhex("dd85b19a")
    assert codec.decode_datavalue(b) == (u, 17)
    out = bytearray()
    codec.encode_datavalue(out, 2095)
    codec.encode_datavalue(out, "MSS")
    a, at = codec.decode_datavalue(bytes(out))
    c, at = codec.decode_datavalue(bytes(out), at)
    assert (a, c, at) == (2095, "MSS", len(out))


def test_encode_decode_datavalues_nested():
    """data/tests/memcmp.rs:115-137"""
    dv = [None, False, True, 1, 1.0, 2 ** 63 - 1, 2 ** 63 - 2, 2 ** 63 - 3, -(2 ** 63), -(2 ** 63) + 1, -(2 ** 63) + 2,
          float("inf"), float("-inf"), []]
    dv.append(list(dv))
    dv.append(list(dv))
    b = enc(dv)
    dec, at = codec.decode_datavalue(b)
    assert at == len(b)

    def same(x, y):
        if isinstance(x, list):
            return isinstance(y, list) and len(x) == len(y) and all(same(p, q) for p, q in zip(x, y))
        return type(x) is type(y) and x == y
    assert same(dec, dv)


def test_tuple_order_is_key_byte_order():
    """the store orders rows by key bytes; the evaluator orders tuples by DataValue -- the two agree"""
    vals = [None, False, True, -5, -5.0, 0, 0.5, 3, 3.0, 10 ** 17, "", "a", "ab", "b", b"", b"\x00", b"z", [], [1], [1, "a"], [2],
            ["a"], [[1]], -(2 ** 60), 2.5e300]
    keyed = sorted(vals, key=enc)
    assert [sort_key(v) for v in keyed] == sorted(sort_key(v) for v in vals)
    rows = [(a, b) for a in vals[:12] for b in vals[8:16]]
    by_bytes = sorted(rows, key=lambda t: codec.encode_key_for_store(7, t))
    by_value = sorted(rows, key=lambda t: tuple(sort_key(x) for x in t))
    assert [tuple(sort_key(x) for x in t) for t in by_bytes] == [tuple(sort_key(x) for x in t) for t in by_value]


def test_vector_key_and_value_forms():
    v = np.array([1.5, -2.0, 0.0, 3.25], dtype=np.float32)
    b = enc(v)
    assert b[:2] == bytes([codec.VEC_TAG, codec.VEC_F32]) and b[2:10] == (4).to_bytes(8, "big")
    assert b[10:14] == np.array([1.5], dtype=">f4").tobytes()  # big-endian elements in keys (memcmp.rs:57-61)
    dec, at = codec.decode_datavalue(b)
    assert at == len(b) and dec.dtype == np.float32 and np.array_equal(dec, v)
    val = codec.encode_val_for_store(3, [v, [v, v], "s", 1, 2.5, None, True, b"\x01\x02"])
    assert val[:8] == (3).to_bytes(8, "big")
    import msgpack
    raw = msgpack.unpackb(val[8:], raw=False)
    assert raw[0] == {"Vec": [0, v.tobytes()]}  # little-endian elements in values (data/value.rs:233-239)
    assert raw[2:] == [{"Str": "s"}, {"Num": {"Int": 1}}, {"Num": {"Float": 2.5}}, "Null", {"Bool": True}, {"Bytes": b"\x01\x02"}]
    tup = codec.decode_tuple_from_kv(codec.encode_key_for_store(3, [9, "k"]), val)
    assert tup[0] == 9 and tup[1] == "k" and np.array_equal(tup[2], v) and np.array_equal(tup[3][1], v)
    assert tup[4:] == ["s", 1, 2.5, None, True, b"\x01\x02"] and type(tup[5]) is int and type(tup[6]) is float


def test_stored_rows_sorted_and_replacing():
    rows = codec.StoredRows.from_tuples(11, [(3, "c", 1.0), (1, "a", 2.0), (2, "b", 3.0), (1, "a", 9.0)], 2)
    assert len(rows) == 3
    assert rows.tuples() == [[1, "a", 9.0], [2, "b", 3.0], [3, "c", 1.0]]
    k, v = rows.row(0)
    assert k[:8] == (11).to_bytes(8, "big") and v[:8] == (11).to_bytes(8, "big")


def test_golden_stored_rows():
    """tests/golden/stored_rows.json (made by tests/golden/make_stored_rows.py): our encoders write today what they wrote when
    the fixture was committed, and the fixture decodes back to the rows"""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_stored_rows", os.path.join(here, "golden", "make_stored_rows.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    golden = json.load(open(os.path.join(here, "golden", "stored_rows.json")))
    assert mod.build() == golden
    g = golden["ints"]
    assert codec.decode_tuple_from_kv(bytes.fromhex(g["key"]), bytes.fromhex(g["val"])) == [2095, -1, 0, 2 ** 53, -(2 ** 63)]
    g = golden["mixed-key-and-value"]
    back = codec.decode_tuple_from_kv(bytes.fromhex(g["key"]), bytes.fromhex(g["val"]))
    assert back == [7, "k", None, True, False, b"\x00\x01", [1, "x", [2.5]], 3.25, "value", -70000, 2 ** 40]
    # the two spelled-out examples of the formats (include/cozo_ingest.h): 2095 as a key column, [5, Null] as a value
    assert codec.memcmp_bytes(2095).hex() == "05" + "c0a05e0000000000" + "00"
    assert codec.encode_val_for_store(3, [5, None]).hex() == "0000000000000003" + "92" + "81a34e756d" + "81a3496e74" + "05" + "a44e756c6c"


def test_stored_rows_delta_is_what_turns_old_into_new():
    """codec.stored_rows_delta: puts = new or changed rows, dels = vanished keys; applying them to `old` gives `new`"""
    rng = np.random.default_rng(4)
    for trial in range(20):
        keys = rng.choice(400, size=int(rng.integers(0, 200)), replace=False)
        old_t = [(int(k), f"v{int(k) % 7}", float(k)) for k in keys]
        keep = [t for t in old_t if rng.random() < 0.8]
        changed = [(t[0], t[1] + "!", t[2]) if rng.random() < 0.2 else t for t in keep]
        fresh = [(int(k), "new", 0.5) for k in rng.choice(np.arange(400, 500), size=int(rng.integers(0, 30)), replace=False)]
        old = codec.StoredRows.from_tuples(9, old_t, 1)
        new = codec.StoredRows.from_tuples(9, changed + fresh, 1)
        puts, dels = codec.stored_rows_delta(old, new)
        assert len(dels) == len(old_t) - len(keep)
        assert len(puts) == len(fresh) + sum(1 for a, b in zip(keep, changed) if a != b)
        got = codec.apply_stored_delta(old, puts, dels)
        assert got.keys == new.keys and got.vals == new.vals and np.array_equal(got.key_off, new.key_off)
    same, none = codec.stored_rows_delta(new, new)
    assert len(same) == 0 and none == []

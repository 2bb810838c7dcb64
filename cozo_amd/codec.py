"""Byte formats of stored rows, host mirror in Python: what the storage iterator hands to cozo-core and what
`store_tx.put` takes back.  Used to write the rows of a GPU-built index back as ordinary `tbl:idx` rows and, in the
tests, to fabricate the stored form of relations for cozo_amd/ingest.py (libcozo_ingest.so is an independent C++
restatement of the same formats; the two are checked against each other).

* key    = 8-byte big-endian relation id + the key columns in the memcmp encoding
           (data/memcmp.rs:46-163 encode, :165-365 decode; data/tuple.rs:27-52; runtime/relation.rs:247-267)
* value  = 8-byte prefix + ONE msgpack array of the non-key columns (runtime/relation.rs:275-296, 526-531) in
           rmp-serde 1.2.0's representation of `enum DataValue` (data/value.rs:143-175): unit variants are strings,
           every other variant a one-entry map {variant name: payload}; Vector is the tuple (0 | 1, bytes of the
           elements in native little-endian order) (data/value.rs:226-252); Bytes goes through serde_bytes (bin).
           rmp-serde is not vendored under /root/reference: this is restated from the crate's published behaviour
           (enum variants by name since 1.0), and the reference's tests hold no stored bytes to pin it against --
           "parity unpinned" for the value side; the C++ decoder therefore also accepts the variant-index form.

Python values stand for DataValues as in cozo_amd/fixed_rule.py: None, bool, int, float, str, bytes, list / tuple,
numpy float32 / float64 1-d arrays (Vec), uuid.UUID."""
from __future__ import annotations

import struct
import uuid
from typing import Any, List, Sequence, Tuple

import numpy as np

INIT_TAG, NULL_TAG, FALSE_TAG, TRUE_TAG, VEC_TAG, NUM_TAG, STR_TAG, BYTES_TAG = 0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07
UUID_TAG, REGEX_TAG, LIST_TAG, SET_TAG, VLD_TAG, JSON_TAG, BOT_TAG = 0x08, 0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0xFF
VEC_F32, VEC_F64 = 0x01, 0x02
IS_FLOAT, IS_APPROX_INT, IS_EXACT_INT = 0b00010000, 0b00000100, 0b00000000
EXACT_INT_BOUND = 0x20_0000_0000_0000
SIGN_MARK = 0x8000000000000000
MASK64 = 0xFFFFFFFFFFFFFFFF
ENCODED_KEY_MIN_LEN = 8  # data/tuple.rs:86


# ---------------------------------------------------------------------------------------------- memcmp
def order_encode_i64(v: int) -> int:  # memcmp.rs:196-198
    return (v & MASK64) ^ SIGN_MARK


def order_decode_i64(u: int) -> int:
    u ^= SIGN_MARK
    return u - (1 << 64) if u >> 63 else u


def order_encode_f64(v: float) -> int:  # memcmp.rs:204-211: sign-positive -> set the top bit, else flip all bits
    u = struct.unpack(">Q", struct.pack(">d", v))[0]
    return (~u & MASK64) if u >> 63 else (u | SIGN_MARK)


def order_decode_f64(u: int) -> float:
    u = (u & ~SIGN_MARK) if u & SIGN_MARK else (~u & MASK64)
    return struct.unpack(">d", struct.pack(">Q", u))[0]


def encode_bytes(out: bytearray, key: bytes) -> None:  # memcmp.rs:147-163
    n = len(key)
    index = 0
    while index <= n:
        remain = n - index
        if remain > 8:
            out += key[index:index + 8]
            out.append(0xFF)
        else:
            pad = 8 - remain
            out += key[index:]
            out += b"\x00" * pad
            out.append(0xFF - pad)
        index += 8


def decode_bytes(data: bytes, at: int = 0) -> Tuple[bytes, int]:  # memcmp.rs:165-192
    key = bytearray()
    while True:
        chunk = data[at:at + 9]
        if len(chunk) < 9:
            raise ValueError("truncated byte-string group")
        at += 9
        pad = 0xFF - chunk[8]
        if pad == 0:
            key += chunk[:8]
            continue
        if pad > 8:
            raise ValueError("bad group marker")
        key += chunk[:8 - pad]
        return bytes(key), at


def encode_num(out: bytearray, v) -> None:  # memcmp.rs:127-145
    if isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)):
        i = int(v)
        if not -(1 << 63) <= i < (1 << 63):
            raise OverflowError("Num::Int is an i64")
        out += struct.pack(">Q", order_encode_f64(float(i)))
        if -EXACT_INT_BOUND < i < EXACT_INT_BOUND:
            out.append(IS_EXACT_INT)
        else:
            out.append(IS_APPROX_INT)
            out += struct.pack(">Q", order_encode_i64(i))
    else:
        out += struct.pack(">Q", order_encode_f64(float(v)))
        out.append(IS_FLOAT)


def decode_num(data: bytes, at: int):  # memcmp.rs:227-245
    f = order_decode_f64(struct.unpack_from(">Q", data, at)[0])
    tag = data[at + 8]
    if tag == IS_FLOAT:
        return f, at + 9
    if tag == IS_EXACT_INT:
        return int(f), at + 9
    if tag == IS_APPROX_INT:
        return order_decode_i64(struct.unpack_from(">Q", data, at + 9)[0]), at + 17
    raise ValueError("bad number kind")


def encode_datavalue(out: bytearray, v: Any) -> None:  # memcmp.rs:47-126
    if v is None:
        out.append(NULL_TAG)
    elif isinstance(v, (bool, np.bool_)):
        out.append(TRUE_TAG if v else FALSE_TAG)
    elif isinstance(v, np.ndarray):
        out.append(VEC_TAG)
        if v.dtype == np.float32:
            out.append(VEC_F32)
            out += struct.pack(">Q", v.size)
            out += v.astype(">f4").tobytes()
        elif v.dtype == np.float64:
            out.append(VEC_F64)
            out += struct.pack(">Q", v.size)
            out += v.astype(">f8").tobytes()
        else:
            raise TypeError("a Vec is f32 or f64")
    elif isinstance(v, (int, float, np.integer, np.floating)):
        out.append(NUM_TAG)
        encode_num(out, v)
    elif isinstance(v, str):
        out.append(STR_TAG)
        encode_bytes(out, v.encode("utf-8"))
    elif isinstance(v, (bytes, bytearray)):
        out.append(BYTES_TAG)
        encode_bytes(out, bytes(v))
    elif isinstance(v, uuid.UUID):
        out.append(UUID_TAG)
        b = v.bytes  # as_fields: d1 = b[0:4], d2 = b[4:6], d3 = b[6:8]; written d3, d2, d1, rest (memcmp.rs:86-93)
        out += b[6:8] + b[4:6] + b[0:4] + b[8:16]
    elif isinstance(v, (list, tuple)):
        out.append(LIST_TAG)
        for el in v:
            encode_datavalue(out, el)
        out.append(INIT_TAG)
    else:
        raise TypeError(f"not a DataValue: {type(v).__name__}")


def decode_datavalue(data: bytes, at: int = 0):  # memcmp.rs:258-365
    tag = data[at]
    at += 1
    if tag == NULL_TAG:
        return None, at
    if tag == FALSE_TAG:
        return False, at
    if tag == TRUE_TAG:
        return True, at
    if tag == NUM_TAG:
        return decode_num(data, at)
    if tag == STR_TAG:
        b, at = decode_bytes(data, at)
        return b.decode("utf-8"), at
    if tag == BYTES_TAG:
        return decode_bytes(data, at)
    if tag == UUID_TAG:
        b = data[at:at + 16]
        return uuid.UUID(bytes=bytes(b[4:8] + b[2:4] + b[0:2] + b[8:16])), at + 16
    if tag == LIST_TAG:
        out: List[Any] = []
        while data[at] != INIT_TAG:
            v, at = decode_datavalue(data, at)
            out.append(v)
        return out, at + 1
    if tag == VEC_TAG:
        t = data[at]
        n = struct.unpack_from(">Q", data, at + 1)[0]
        at += 9
        if t == VEC_F32:
            return np.frombuffer(data, dtype=">f4", count=n, offset=at).astype(np.float32), at + 4 * n
        if t == VEC_F64:
            return np.frombuffer(data, dtype=">f8", count=n, offset=at).astype(np.float64), at + 8 * n
        raise ValueError("bad vector element tag")
    raise ValueError(f"unsupported key tag 0x{tag:02x}")


def memcmp_bytes(v: Any) -> bytes:
    out = bytearray()
    encode_datavalue(out, v)
    return bytes(out)


def encode_key_for_store(relation_id: int, key_columns: Sequence[Any]) -> bytes:
    """RelationHandle::encode_key_for_store (runtime/relation.rs:247-267) = Tuple::encode_as_key (data/tuple.rs:27-38)"""
    out = bytearray(struct.pack(">Q", relation_id))
    for v in key_columns:
        encode_datavalue(out, v)
    return bytes(out)


def decode_tuple_from_key(key: bytes) -> List[Any]:  # data/tuple.rs:41-52
    at = ENCODED_KEY_MIN_LEN
    out = []
    while at < len(key):
        v, at = decode_datavalue(key, at)
        out.append(v)
    return out


# ---------------------------------------------------------------------------------------------- msgpack values
def _to_serde(v: Any):
    """the msgpack-able shape of one DataValue as rmp-serde 1.2.0 writes the derived enum"""
    if v is None:
        return "Null"
    if isinstance(v, (bool, np.bool_)):
        return {"Bool": bool(v)}
    if isinstance(v, np.ndarray):
        if v.dtype == np.float32:
            return {"Vec": [0, v.astype("<f4").tobytes()]}
        if v.dtype == np.float64:
            return {"Vec": [1, v.astype("<f8").tobytes()]}
        raise TypeError("a Vec is f32 or f64")
    if isinstance(v, (int, np.integer)):
        return {"Num": {"Int": int(v)}}
    if isinstance(v, (float, np.floating)):
        return {"Num": {"Float": float(v)}}
    if isinstance(v, str):
        return {"Str": v}
    if isinstance(v, (bytes, bytearray)):
        return {"Bytes": bytes(v)}
    if isinstance(v, uuid.UUID):
        return {"Uuid": v.bytes}
    if isinstance(v, (list, tuple)):
        return {"List": [_to_serde(x) for x in v]}
    raise TypeError(f"not a DataValue: {type(v).__name__}")


_VARIANTS = ["Null", "Bool", "Num", "Str", "Bytes", "Uuid", "Regex", "List", "Set", "Vec", "Json", "Validity", "Bot"]


def _from_serde(o: Any):
    if isinstance(o, (str, int)) and not isinstance(o, bool):
        name = _VARIANTS[o] if isinstance(o, int) else o
        if name == "Null":
            return None
        raise ValueError(f"unsupported unit variant {name}")
    (name, payload), = o.items()
    if isinstance(name, int):
        name = _VARIANTS[name]
    if name == "Bool":
        return bool(payload)
    if name == "Num":
        (kind, x), = payload.items()
        return int(x) if kind in ("Int", 0) else float(x)
    if name == "Str":
        return payload
    if name == "Bytes":
        return bytes(payload)
    if name == "Uuid":
        return uuid.UUID(bytes=bytes(payload))
    if name == "List":
        return [_from_serde(x) for x in payload]
    if name == "Vec":
        el, b = payload
        return np.frombuffer(bytes(b), dtype="<f4" if el == 0 else "<f8").astype(np.float32 if el == 0 else np.float64)
    raise ValueError(f"unsupported variant {name}")


def encode_val_for_store(relation_id: int, value_columns: Sequence[Any]) -> bytes:
    """RelationHandle::encode_val_for_store / encode_val_only_for_store (runtime/relation.rs:275-296)"""
    import msgpack
    return struct.pack(">Q", relation_id) + msgpack.packb([_to_serde(v) for v in value_columns], use_bin_type=True)


def extend_tuple_from_v(tup: List[Any], val: bytes) -> None:  # runtime/relation.rs:526-531
    import msgpack
    if val:
        tup.extend(_from_serde(x) for x in msgpack.unpackb(val[ENCODED_KEY_MIN_LEN:], raw=False, strict_map_key=False))


def decode_tuple_from_kv(key: bytes, val: bytes) -> List[Any]:  # runtime/relation.rs:520-524
    tup = decode_tuple_from_key(key)
    extend_tuple_from_v(tup, val)
    return tup


class StoredRows:
    """The rows of one relation as a scan yields them: concatenated key bytes / value bytes + offsets, ascending by
    key.  `from_tuples` is the write path (what a sequence of store_tx.put calls leaves behind)."""

    def __init__(self, keys: bytes, key_off: np.ndarray, vals: bytes, val_off: np.ndarray, n_key_cols: int):
        self.keys, self.key_off, self.vals, self.val_off, self.n_key_cols = keys, key_off, vals, val_off, n_key_cols

    def __len__(self):
        return self.key_off.size - 1

    @classmethod
    def from_tuples(cls, relation_id: int, tuples: Sequence[Sequence[Any]], n_key_cols: int) -> "StoredRows":
        kv = {}
        for t in tuples:  # a later put of the same key replaces the earlier one
            kv[encode_key_for_store(relation_id, t[:n_key_cols])] = encode_val_for_store(relation_id, t[n_key_cols:])
        items = sorted(kv.items())  # the store orders rows by key BYTES
        key_off = np.zeros(len(items) + 1, dtype=np.uint64)
        val_off = np.zeros(len(items) + 1, dtype=np.uint64)
        key_off[1:] = np.cumsum([len(k) for k, _ in items], dtype=np.uint64) if items else []
        val_off[1:] = np.cumsum([len(v) for _, v in items], dtype=np.uint64) if items else []
        return cls(b"".join(k for k, _ in items), key_off, b"".join(v for _, v in items), val_off, n_key_cols)

    def row(self, i: int) -> Tuple[bytes, bytes]:
        return (self.keys[int(self.key_off[i]):int(self.key_off[i + 1])],
                self.vals[int(self.val_off[i]):int(self.val_off[i + 1])])

    def tuples(self) -> List[List[Any]]:
        return [decode_tuple_from_kv(*self.row(i)) for i in range(len(self))]

    @classmethod
    def from_items(cls, items: Sequence[Tuple[bytes, bytes]], n_key_cols: int) -> "StoredRows":
        """(key bytes, value bytes) pairs, ascending by key"""
        key_off = np.zeros(len(items) + 1, dtype=np.uint64)
        val_off = np.zeros(len(items) + 1, dtype=np.uint64)
        if items:
            key_off[1:] = np.cumsum([len(k) for k, _ in items], dtype=np.uint64)
            val_off[1:] = np.cumsum([len(v) for _, v in items], dtype=np.uint64)
        return cls(b"".join(k for k, _ in items), key_off, b"".join(v for _, v in items), val_off, n_key_cols)


def stored_rows_delta(old: StoredRows, new: StoredRows) -> Tuple[StoredRows, List[bytes]]:
    """What a statement has to write to turn the stored rows `old` into `new`: (puts, dels) -- the rows of `new` whose key is
    not in `old` or whose value bytes differ, as StoredRows ready for store_tx.put, and the keys of `old` that `new` no longer
    has, for store_tx.del.  This is the write-back of index maintenance on the device: after
    cz_hnsw_insert / cz_hnsw_remove the index's `tbl:idx` rows are encoded again (GpuHnswIndex.index_rows) and only the delta
    against the rows the store holds is written -- the rows hnsw_put_vector / hnsw_remove_vec would have touched
    (runtime/hnsw.rs:270-357, 389-466, 728-868).  Both inputs ascending by key bytes (a scan's order): one merge walk."""
    puts: List[Tuple[bytes, bytes]] = []
    dels: List[bytes] = []
    i, j, n_old, n_new = 0, 0, len(old), len(new)
    while i < n_old or j < n_new:
        if j == n_new:
            dels.append(old.row(i)[0])
            i += 1
            continue
        if i == n_old:
            puts.append(new.row(j))
            j += 1
            continue
        (ko, vo), (kn, vn) = old.row(i), new.row(j)
        if ko == kn:
            if vo != vn:
                puts.append((kn, vn))
            i += 1
            j += 1
        elif ko < kn:
            dels.append(ko)
            i += 1
        else:
            puts.append((kn, vn))
            j += 1
    return StoredRows.from_items(puts, new.n_key_cols), dels


def apply_stored_delta(rows: StoredRows, puts: StoredRows, dels: Sequence[bytes]) -> StoredRows:
    """the store after `dels` and `puts` (tests: delta(old, new) applied to old is new)"""
    kv = {rows.row(i)[0]: rows.row(i)[1] for i in range(len(rows))}
    for k in dels:
        kv.pop(k, None)
    for i in range(len(puts)):
        k, v = puts.row(i)
        kv[k] = v
    return StoredRows.from_items(sorted(kv.items()), rows.n_key_cols)

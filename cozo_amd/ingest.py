"""ctypes binding of libcozo_ingest.so (include/cozo_ingest.h): stored rows -> the flat arrays libcozo_gpu.so takes,
and the way back (a flat index -> the tuples of its `tbl:idx` relation).

Host code only: loads and runs without a GPU.  The stored-row formats are documented in cozo_amd/codec.py."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
from typing import Any, Callable, List, Optional, Sequence

import numpy as np

from . import codec
from ._lib import CZ_NONE, HnswDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("COZO_INGEST_LIB") or os.path.join(_HERE, "lib", "libcozo_ingest.so")

CZI_OK, CZI_E_INVALID, CZI_E_CORRUPT, CZI_E_NOT_AN_EDGE, CZI_E_BAD_WEIGHT = 0, -1, -2, -3, -4
CZI_E_UNSUPPORTED, CZI_E_TOO_LARGE, CZI_E_MISSING_ROW, CZI_E_OOM = -5, -6, -7, -8
CZI_UNDIRECTED, CZI_WEIGHTED, CZI_ALLOW_NEGATIVE_WEIGHTS, CZI_ORDERED_IDS = 1, 2, 4, 8

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


class Rows(C.Structure):  # czi_rows
    _fields_ = [("keys", C.c_void_p), ("key_off", C.c_void_p), ("vals", C.c_void_p), ("val_off", C.c_void_p),
                ("n_rows", C.c_uint64), ("n_key_cols", C.c_uint32)]


class CozoIngestError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcozo_ingest error {code}: {msg}")
        self.code = code


# every symbol include/cozo_ingest.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "czi_last_error": (C.c_char_p, []),
    "czi_version": (C.c_char_p, []),
    "czi_graph_ingest": (C.c_int, [C.POINTER(Rows), C.c_uint32, C.POINTER(C.c_void_p)]),
    "czi_graph_free": (None, [C.c_void_p]),
    "czi_graph_node_count": (C.c_uint32, [C.c_void_p]),
    "czi_graph_edge_count": (C.c_uint64, [C.c_void_p]),
    "czi_graph_csr": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "czi_graph_node_keys": (C.c_int, [C.c_void_p, C.POINTER(u8p), C.POINTER(u64p)]),
    "czi_graph_lookup": (C.c_uint32, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "czi_hnsw_ingest": (C.c_int, [C.POINTER(Rows), C.POINTER(Rows), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32,
                                  C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "czi_hnsw_ingest_f64": (C.c_int, [C.POINTER(Rows), C.POINTER(Rows), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32,
                                  C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "czi_hnsw_free": (None, [C.c_void_p]),
    "czi_hnsw_desc": (C.c_int, [C.c_void_p, C.POINTER(HnswDesc), C.POINTER(f32p)]),
    "czi_hnsw_desc_f64": (C.c_int, [C.c_void_p, C.POINTER(HnswDesc), C.POINTER(C.POINTER(C.c_double))]),
    "czi_hnsw_nodes": (C.c_int, [C.c_void_p, C.POINTER(u64p), C.POINTER(u32p), C.POINTER(i32p)]),
    "czi_hnsw_row_counts": (C.c_int, [C.c_void_p, u64p, u64p, u64p, u64p]),
    "czi_hnsw_encode_rows_degrees": (C.c_int, [C.POINTER(HnswDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_uint64, C.POINTER(C.c_void_p)]),
    "czi_hnsw_encode_rows": (C.c_int, [C.POINTER(HnswDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.POINTER(C.c_void_p)]),
    "czi_row_buf_rows": (C.c_int, [C.c_void_p, C.POINTER(Rows)]),
    "czi_row_buf_free": (None, [C.c_void_p]),
}

_L = None


def lib() -> C.CDLL:
    global _L
    if _L is None:
        if not os.path.exists(SO_PATH):
            raise OSError(f"{SO_PATH} is missing: run `python -m cozo_amd.build` (or __graft_entry__.build())")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _L = L
    return _L


def check(rc: int) -> None:
    if rc != CZI_OK:
        raise CozoIngestError(rc, lib().czi_last_error().decode("utf-8", "replace"))


class _RowsArg:
    """keeps the buffers of a codec.StoredRows alive for the duration of a call"""

    def __init__(self, rows: codec.StoredRows):
        self.keys = np.frombuffer(rows.keys, dtype=np.uint8) if len(rows.keys) else np.zeros(1, np.uint8)
        self.vals = np.frombuffer(rows.vals, dtype=np.uint8) if len(rows.vals) else np.zeros(1, np.uint8)
        self.key_off = np.ascontiguousarray(rows.key_off, dtype=np.uint64)
        self.val_off = np.ascontiguousarray(rows.val_off, dtype=np.uint64)
        self.c = Rows(self.keys.ctypes.data, self.key_off.ctypes.data, self.vals.ctypes.data, self.val_off.ctypes.data,
                      len(rows), rows.n_key_cols)


class StoredGraph:
    """FixedRuleInputRelation::as_directed_graph / as_directed_weighted_graph (fixed_rule/mod.rs:136-328) over the
    stored bytes of a relation."""

    def __init__(self, rows: codec.StoredRows, undirected: bool = False, weighted: bool = False,
                 allow_negative_weights: bool = False, ordered_ids: bool = False):
        """ordered_ids: ids = rank of the node value (what ShortestPathBFS / Bfs need, as_ordered_graph in
        cozo_amd/fixed_rule.py) instead of first appearance"""
        arg = _RowsArg(rows)
        h = C.c_void_p()
        flags = (CZI_UNDIRECTED if undirected else 0) | (CZI_WEIGHTED if weighted else 0) | \
            (CZI_ALLOW_NEGATIVE_WEIGHTS if allow_negative_weights else 0) | (CZI_ORDERED_IDS if ordered_ids else 0)
        check(lib().czi_graph_ingest(C.byref(arg.c), flags, C.byref(h)))
        self._h = h
        self.weighted = weighted
        self.n = int(lib().czi_graph_node_count(h))
        self.e = int(lib().czi_graph_edge_count(h))

    def close(self):
        if self._h:
            lib().czi_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def csr(self, inverse: bool = False):
        """(offsets u32 [N+1], targets u32 [E], weights f32 [E] or None)"""
        off = np.empty(self.n + 1, dtype=np.uint32)
        tgt = np.empty(self.e, dtype=np.uint32)
        w = np.empty(self.e, dtype=np.float32) if self.weighted else None
        check(lib().czi_graph_csr(self._h, int(inverse), off.ctypes.data, tgt.ctypes.data, None if w is None else w.ctypes.data))
        return off, tgt, w

    def node_keys(self):
        """(concatenated memcmp bytes, offsets [N+1]) of the node values in id order"""
        b, o = u8p(), u64p()
        check(lib().czi_graph_node_keys(self._h, C.byref(b), C.byref(o)))
        off = np.ctypeslib.as_array(o, shape=(self.n + 1,)).copy()
        data = bytes(np.ctypeslib.as_array(b, shape=(int(off[-1]),))) if off[-1] else b""
        return data, off

    def indices(self) -> List[Any]:
        """the reference's `indices`: id -> node value (N decodes, not 2E)"""
        data, off = self.node_keys()
        return [codec.decode_datavalue(data, int(off[i]))[0] for i in range(self.n)]

    def get_node_idx(self, value: Any) -> Optional[int]:
        k = codec.memcmp_bytes(value)
        i = int(lib().czi_graph_lookup(self._h, k, len(k)))
        return None if i == CZ_NONE else i


class StoredHnswIndex:
    """The flat form of one `tbl:idx` relation + the indexed vectors of its base relation, read off the stored bytes."""

    def __init__(self, idx: codec.StoredRows, base: codec.StoredRows, vec_fields: Sequence[int], dim: int, metric: int,
                 m_neighbours: int, dtype: str = "F32"):
        """dtype: the manifest's VecElementType (runtime/hnsw.rs:33-44); an F64 index keeps its vectors as f64 (czi_hnsw_ingest_f64)"""
        a, b = _RowsArg(idx), _RowsArg(base)
        vf = np.ascontiguousarray(vec_fields, dtype=np.uint32)
        h = C.c_void_p()
        f64 = dtype == "F64"
        ingest = lib().czi_hnsw_ingest_f64 if f64 else lib().czi_hnsw_ingest
        check(ingest(C.byref(a.c), C.byref(b.c), vf.ctypes.data, vf.size, dim, metric, m_neighbours, 2 * m_neighbours, C.byref(h)))
        self._h = h
        self.dtype = "F64" if f64 else "F32"
        d, v = HnswDesc(), (C.POINTER(C.c_double)() if f64 else f32p())
        check((lib().czi_hnsw_desc_f64 if f64 else lib().czi_hnsw_desc)(h, C.byref(d), C.byref(v)))
        self.n, self.dim, self.metric, self.n_levels, self.entry = d.n, d.dim, d.metric, d.n_levels, d.entry
        self.vectors = np.ctypeslib.as_array(v, shape=(d.n, d.dim)).copy() if d.n else np.zeros((0, dim), np.float64 if f64 else np.float32)
        self.level_size = [int(d.level_size[l]) for l in range(d.n_levels)]
        self.level_width = [int(d.level_width[l]) for l in range(d.n_levels)]
        self.level_nodes = [np.ctypeslib.as_array(d.level_nodes[l], shape=(self.level_size[l],)).copy()
                            for l in range(d.n_levels)]
        self.level_nbrs = [np.ctypeslib.as_array(d.level_nbrs[l], shape=(self.level_size[l], self.level_width[l])).copy()
                           for l in range(d.n_levels)]
        br, fl, sb = u64p(), u32p(), i32p()
        check(lib().czi_hnsw_nodes(h, C.byref(br), C.byref(fl), C.byref(sb)))
        if d.n:
            self.nodes = list(zip(np.ctypeslib.as_array(br, shape=(d.n,)).tolist(),
                                  np.ctypeslib.as_array(fl, shape=(d.n,)).tolist(),
                                  np.ctypeslib.as_array(sb, shape=(d.n,)).tolist()))
        else:
            self.nodes = []
        c = (C.c_uint64 * 4)()
        check(lib().czi_hnsw_row_counts(h, *(C.cast(C.byref(c, 8 * i), u64p) for i in range(4))))
        self.n_rows, self.n_self, self.n_live_links, self.n_ignored = (int(x) for x in c)
        lib().czi_hnsw_free(h)
        self._h = None

    def to_gpu(self, manifest):
        """upload: the index handle cz_hnsw_search_batch takes (cozo_amd.hnsw.GpuHnswIndex)"""
        from .hnsw import GpuHnswIndex
        return GpuHnswIndex(manifest, self.vectors, [None] + self.level_nodes[1:], self.level_nbrs, self.entry)


def index_relation_tuples(key_of_node: Sequence[Sequence[Any]], vectors: np.ndarray, level_nodes: Sequence, level_nbrs: Sequence,
                          entry: int, link_distance: Callable[[np.ndarray], np.ndarray], relation_id: int = 0) -> List[tuple]:
    """The way back: the tuples of the `tbl:idx` relation (schema runtime/relation.rs:1064-1126)
    that hold a flat index -- what hnsw_put_vector leaves in the store (runtime/hnsw.rs:270-330, 630-678):
      * per level and node the self-loop row [layer, fr.., fr..] -> [degree as f64, SHA-256 of the vector's
        little-endian bytes (data/value.rs:333-348), false]
      * per live link the row [layer, fr.., to..] -> [distance f64, Null, false]
      * the canary row [1, Null..] -> [bottom layer, key bytes of the entry's self row with a Null layer, false] (:641-669)
    key_of_node[i] = the CompoundKey of node i flattened: (row key columns.., field, sub index).
    level_nodes[l] may be None on level 0 (identity).  link_distance(pairs [P][2]) -> f64 [P] (the GPU build binds it to
    cozo_amd.hnsw.distance_batch, so the stored distances are the ones the kernels computed)."""
    rows: List[tuple] = []
    n_levels = len(level_nbrs)
    if n_levels == 0:
        return rows
    width = len(key_of_node[0])
    hashes = {}
    for lv in range(n_levels):
        ids = np.arange(len(level_nbrs[lv])) if level_nodes[lv] is None else np.asarray(level_nodes[lv])
        tab = np.asarray(level_nbrs[lv])
        pairs = [(int(ids[r]), int(t)) for r in range(tab.shape[0]) for t in tab[r] if t != CZ_NONE]
        d = link_distance(np.asarray(pairs, dtype=np.uint32).reshape(-1, 2)) if pairs else np.zeros(0)
        at = 0
        for r in range(tab.shape[0]):
            fr = int(ids[r])
            live = [int(t) for t in tab[r] if t != CZ_NONE]
            if fr not in hashes:
                hashes[fr] = hashlib.sha256(np.ascontiguousarray(vectors[fr], dtype="<f4").tobytes()).digest()
            rows.append((-lv, *key_of_node[fr], *key_of_node[fr], float(len(live)), hashes[fr], False))
            for t in live:
                rows.append((-lv, *key_of_node[fr], *key_of_node[t], float(d[at]), None, False))
                at += 1
    target_key = codec.encode_key_for_store(relation_id, (None, *key_of_node[entry], *key_of_node[entry]))
    rows.append((1, *([None] * (2 * width)), -(n_levels - 1), target_key, False))
    return rows


def encode_index_rows(key_of_node: Sequence[Sequence[Any]], vectors: np.ndarray, level_nodes: Sequence, level_nbrs: Sequence,
                      entry: int, metric: int, level_dist: Sequence[np.ndarray], relation_id: int,
                      level_degree: Optional[Sequence[np.ndarray]] = None) -> codec.StoredRows:
    """index_relation_tuples + the store's encoding in one native pass (czi_hnsw_encode_rows_degrees): the key / value bytes of
    every `tbl:idx` row of a flat index, in key order.  level_dist[l] [size][width] f64 = the distance of every link slot;
    level_degree[l] [size] f64 = the degree of every self row (None: the number of link rows)."""
    from ._lib import i32p as _i32p, u32p as _u32p
    vectors = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = vectors.shape
    L = len(level_nbrs)
    sizes = np.array([len(t) for t in level_nbrs], dtype=np.uint32)
    widths = np.array([np.asarray(t).shape[1] for t in level_nbrs], dtype=np.int32)
    nodes = [None if level_nodes[l] is None else np.ascontiguousarray(level_nodes[l], dtype=np.uint32) for l in range(L)]
    nbrs = [np.ascontiguousarray(level_nbrs[l], dtype=np.uint32) for l in range(L)]
    dists = [np.ascontiguousarray(level_dist[l], dtype=np.float64) for l in range(L)]
    nodes_p = (_u32p * max(L, 1))(*[C.cast(None if a is None else a.ctypes.data, _u32p) for a in nodes])
    nbrs_p = (_u32p * max(L, 1))(*[C.cast(a.ctypes.data, _u32p) for a in nbrs])
    dist_p = (C.c_void_p * max(L, 1))(*[a.ctypes.data for a in dists])
    d = HnswDesc(n, dim, metric, L, entry, C.cast(sizes.ctypes.data, _u32p), C.cast(widths.ctypes.data, _i32p), nodes_p, nbrs_p)
    keys = [b"".join(codec.memcmp_bytes(v) for v in k) for k in key_of_node]
    off = np.zeros(len(keys) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(k) for k in keys], dtype=np.uint64) if keys else []
    blob = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8)
    h = C.c_void_p()
    degs = None if level_degree is None else [np.ascontiguousarray(level_degree[l], dtype=np.float64) for l in range(L)]
    deg_p = None if degs is None else (C.c_void_p * max(L, 1))(*[a.ctypes.data for a in degs])
    check(lib().czi_hnsw_encode_rows_degrees(C.byref(d), vectors.ctypes.data, blob.ctypes.data, off.ctypes.data, dist_p, deg_p,
                                             relation_id, C.byref(h)))
    r = Rows()
    check(lib().czi_row_buf_rows(h, C.byref(r)))
    nr = int(r.n_rows)
    key_off = np.ctypeslib.as_array(C.cast(r.key_off, u64p), shape=(nr + 1,)).copy()
    val_off = np.ctypeslib.as_array(C.cast(r.val_off, u64p), shape=(nr + 1,)).copy()
    kb = C.string_at(r.keys, int(key_off[-1])) if nr else b""
    vb = C.string_at(r.vals, int(val_off[-1])) if nr else b""
    lib().czi_row_buf_free(h)
    width = len(key_of_node[0]) if len(key_of_node) else 0
    return codec.StoredRows(kb, key_off, vb, val_off, 2 * width + 1)

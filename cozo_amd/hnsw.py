"""Host-side mirror of the reference's HNSW operator surface over libcozo_gpu.

Names follow cozo-core: `HnswIndexManifest` (runtime/hnsw.rs:27-43), `HnswSearch`
(data/program.rs:975-991), `hnsw_knn` (runtime/hnsw.rs:869-1012).  The reference runs `hnsw_knn` once per
parent tuple inside `HnswSearchRA::iter` (query/ra.rs:1085-1121); `hnsw_knn_batch` is the same operator
applied to the whole batch of parent tuples in one GPU launch, results re-emitted in parent order.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import CZ_COSINE, CZ_DEVICE_PTRS, CZ_IP, CZ_L2, CZ_NONE, check, ptr

DISTANCES = {"L2": CZ_L2, "Cosine": CZ_COSINE, "IP": CZ_IP}  # HnswDistance, parse/sys.rs:76-98


@dataclass
class HnswIndexManifest:
    """runtime/hnsw.rs:27-43 (the fields the search path reads)."""
    vec_dim: int
    distance: str = "L2"
    m_neighbours: int = 16
    ef_construction: int = 100
    dtype: str = "F32"
    extend_candidates: bool = False
    keep_pruned_connections: bool = False

    @property
    def m_max(self) -> int:  # runtime/relation.rs:1136-1151
        return self.m_neighbours

    @property
    def m_max0(self) -> int:
        return self.m_neighbours * 2

    @property
    def level_multiplier(self) -> float:
        return 1.0 / float(np.log(self.m_neighbours))


@dataclass
class HnswSearch:
    """data/program.rs:975-991: per-query search parameters."""
    k: int
    ef: int
    radius: Optional[float] = None
    has_filter: bool = False  # a filter keeps all ef candidates until after filtering (hnsw.rs:943-947)


def _build_flags(manifest) -> int:
    return _lib.CZ_HNSW_EXTEND_CANDIDATES if manifest.extend_candidates else 0


def _shared_rows(row_of) -> bool:
    """does any base row carry more than one indexed vector (a List of vectors, several vec_fields: hnsw.rs:694-706)?"""
    if row_of is None:
        return False
    r = np.asarray(row_of)
    return bool(r.size and np.unique(r).size != r.size)


class GpuHnswIndex:
    """A device-resident flat export of one `tbl:idx` relation (see include/cozo_gpu.h cz_hnsw_desc)."""

    def __init__(self, manifest: HnswIndexManifest, vectors: np.ndarray, level_nodes: Sequence[np.ndarray],
                 level_nbrs: Sequence[np.ndarray], entry: int):
        if manifest.dtype not in ("F32", "F64"):  # VecElementType, parse/sys.rs
            raise _lib.CozoGpuError(_lib.CZ_E_UNSUPPORTED, f"vector element type {manifest.dtype!r}")
        self.manifest = manifest
        # an F64 index keeps f64 vectors and every distance is computed in f64 (VectorCache::dist's F64 arms, hnsw.rs:73-78,
        # 86-95, 102-106): searched on the device, built / maintained on the reference's CPU path
        self._np = np.float64 if manifest.dtype == "F64" else np.float32
        vectors = np.ascontiguousarray(vectors, dtype=self._np)
        if vectors.ndim != 2 or vectors.shape[1] != manifest.vec_dim:
            raise ValueError("vectors must be [n][vec_dim]")
        self.n = vectors.shape[0]
        nl = len(level_nbrs)
        nodes = [None if x is None else np.ascontiguousarray(x, dtype=np.uint32) for x in level_nodes]
        nbrs = [np.ascontiguousarray(x, dtype=np.uint32) for x in level_nbrs]
        size = np.array([x.shape[0] for x in nbrs], dtype=np.uint32)
        width = np.array([x.shape[1] for x in nbrs], dtype=np.int32)
        node_ptrs = (_lib.u32p * max(nl, 1))(*[None if x is None else x.ctypes.data_as(_lib.u32p) for x in nodes])
        nbr_ptrs = (_lib.u32p * max(nl, 1))(*[x.ctypes.data_as(_lib.u32p) for x in nbrs])
        desc = _lib.HnswDesc(self.n, manifest.vec_dim, DISTANCES[manifest.distance], nl, int(entry) & 0xFFFFFFFF,
                             size.ctypes.data_as(_lib.u32p), width.ctypes.data_as(_lib.i32p), node_ptrs, nbr_ptrs)
        h = C.c_void_p()
        create = _lib.lib().cz_hnsw_index_create_f64 if manifest.dtype == "F64" else _lib.lib().cz_hnsw_index_create
        check(create(C.byref(desc), ptr(vectors), C.byref(h)))
        self._h = h

    @classmethod
    def build(cls, manifest: HnswIndexManifest, vectors, levels: Optional[np.ndarray] = None, seed: int = 0,
              max_batch: int = 0, device_ptr: bool = False, n: Optional[int] = None, stream: int = 0, row_of=None):
        """`::hnsw create` on the GPU (create_hnsw_index, runtime/relation.rs:1010-1201 -> hnsw_put per row):
        batch-parallel insertion of all vectors in key order.  `vectors` is a host array, or (device_ptr=True) a
        device tensor / pointer with `n` rows.  Returns the index; `.last_build_n_dist` holds the distance count.
        row_of: the base row each vector comes from (index_nodes gives it).  hnsw_get_neighbours drops every link inside one
        base row (hnsw.rs:609-610): such links are written and counted into the degrees but never read.  When some row carries
        several vectors the index is built as empty handle + cz_hnsw_set_row_of + cz_hnsw_insert, and the tables hold no link
        inside a row (`degrees()` counts them)."""
        if manifest.dtype != "F32":
            raise _lib.CozoGpuError(_lib.CZ_E_UNSUPPORTED, "only F32 vector indices are GPU-resident")
        if _shared_rows(row_of):
            if device_ptr:
                raise ValueError("rows carrying several vectors: hand the vectors in as a host array")
            self = cls.__new__(cls)
            self.manifest = manifest
            h = C.c_void_p()
            check(_lib.lib().cz_hnsw_build(None, 0, manifest.vec_dim, DISTANCES[manifest.distance], manifest.m_neighbours,
                                           manifest.ef_construction, int(manifest.keep_pruned_connections), None, 0, 0, None,
                                           C.byref(h), 0, None))
            self._h = h
            self.n = 0
            self.last_build_n_dist = 0
            self.insert(vectors, levels=levels, seed=seed, max_batch=max_batch, row_of=row_of)
            return self
        self = cls.__new__(cls)
        self.manifest = manifest
        if device_ptr:
            nn = int(n if n is not None else vectors.shape[0])
            vp = ptr(vectors)
        else:
            vectors = np.ascontiguousarray(vectors, dtype=np.float32)
            nn = vectors.shape[0]
            vp = ptr(vectors)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.int32)
        nd = C.c_uint64(0)
        h = C.c_void_p()
        check(_lib.lib().cz_hnsw_build(vp, nn, manifest.vec_dim, DISTANCES[manifest.distance], manifest.m_neighbours,
                                       manifest.ef_construction, int(manifest.keep_pruned_connections), ptr(lv),
                                       int(seed), int(max_batch), C.byref(nd), C.byref(h),
                                       (CZ_DEVICE_PTRS if device_ptr else 0) | _build_flags(manifest), C.c_void_p(stream)))
        self._h = h
        self.n = nn
        self.last_build_n_dist = nd.value
        return self

    def set_key_order(self, key_rank):
        """key_rank[node] = position of the node's (row key, field, sub-index) among all of them, for the nodes held and the
        ones the next insert adds (cz_hnsw_set_key_order); None = ids are in key order.  The reference's entry point is
        the smallest KEY on the top layer (hnsw.rs:184-191, 891-899), which insert / remove then reproduce."""
        if key_rank is None:
            check(_lib.lib().cz_hnsw_set_key_order(self._h, None, 0))
            return
        r = np.ascontiguousarray(key_rank, dtype=np.uint32)
        check(_lib.lib().cz_hnsw_set_key_order(self._h, ptr(r), r.size))

    def insert(self, vectors, levels: Optional[np.ndarray] = None, seed: int = 0, max_batch: int = 0, key_rank=None, row_of=None):
        """hnsw_put for more rows on a later write (stored.rs:431-450 -> hnsw.rs:679-727): the vectors become nodes
        n .. n + len - 1 of this index (cz_hnsw_insert).  key_rank: see set_key_order -- needed when the new rows' keys do
        not all sort behind the existing ones.  row_of: the base row of every node's vector, the nodes held and the ones
        inserted (cz_hnsw_set_row_of; see build) -- needed as soon as some row carries several vectors."""
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.manifest.vec_dim:
            raise ValueError("vectors must be [n][vec_dim]")
        if row_of is not None:
            r = np.ascontiguousarray(row_of, dtype=np.uint32)
            if r.size != self.n + v.shape[0]:
                raise ValueError("row_of must cover the nodes held and the ones inserted")
            check(_lib.lib().cz_hnsw_set_row_of(self._h, ptr(r), r.size))
        if key_rank is not None:
            if len(key_rank) != self.n + v.shape[0]:
                raise ValueError("key_rank must cover the nodes held and the ones inserted")
            self.set_key_order(key_rank)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.int32)
        nd = C.c_uint64(0)
        man = self.manifest
        check(_lib.lib().cz_hnsw_insert(self._h, ptr(v), v.shape[0], man.m_neighbours, man.ef_construction,
                                        int(man.keep_pruned_connections), ptr(lv), int(seed), int(max_batch), C.byref(nd),
                                        _build_flags(man), None))
        self.n += v.shape[0]
        self.last_build_n_dist = nd.value

    def remove(self, nodes):
        """hnsw_remove (hnsw.rs:728-868) for a set of nodes: they leave every level, links from and to them disappear"""
        a = np.ascontiguousarray(nodes, dtype=np.uint32)
        check(_lib.lib().cz_hnsw_remove(self._h, ptr(a), a.size))
        self.__dict__.setdefault("_removed", set()).update(int(x) for x in a)  # (a removed node keeps its id; it has no rows)

    def degrees(self):
        """per level the f64 of every self row (hnsw.rs:270, 338-357) in export()'s node order: the number of link rows, plus
        one where an extend_candidates shrink selected the node itself (cz_hnsw_index_export_degrees)"""
        L = _lib.lib()
        n, dim, metric, nl, entry = C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_int32(), C.c_uint32()
        check(L.cz_hnsw_index_info(self._h, C.byref(n), C.byref(dim), C.byref(metric), C.byref(nl), C.byref(entry)))
        out = []
        for lv in range(nl.value):
            size, width = C.c_uint32(), C.c_int32()
            check(L.cz_hnsw_index_level_info(self._h, lv, C.byref(size), C.byref(width)))
            d = np.empty(size.value, dtype=np.float64)
            check(L.cz_hnsw_index_export_degrees(self._h, lv, ptr(d)))
            out.append(d)
        return out

    def export(self):
        """(level_nodes, level_nbrs, entry): the flat layout of cz_hnsw_desc, e.g. to write the links back as
        `tbl:idx` rows or to hand the same index to another searcher."""
        L = _lib.lib()
        n, dim, metric, nl, entry = C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_int32(), C.c_uint32()
        check(L.cz_hnsw_index_info(self._h, C.byref(n), C.byref(dim), C.byref(metric), C.byref(nl), C.byref(entry)))
        nodes, nbrs = [], []
        for lv in range(nl.value):
            size, width = C.c_uint32(), C.c_int32()
            check(L.cz_hnsw_index_level_info(self._h, lv, C.byref(size), C.byref(width)))
            ids = np.empty(size.value, dtype=np.uint32)
            tab = np.empty((size.value, width.value), dtype=np.uint32)
            check(L.cz_hnsw_index_export_level(self._h, lv, ptr(ids), ptr(tab)))
            nodes.append(ids)
            nbrs.append(tab)
        return nodes, nbrs, entry.value

    def index_rows(self, key_of_node: Sequence[Sequence], relation_id: int):
        """The way back: every `tbl:idx` row of this index as stored key / value bytes in key order
        (cozo_amd.codec.StoredRows), ready for store_tx.put -- link tables exported from the device, link distances from
        cz_distance_batch (the values the kernels work with), self-loop rows with degree and vector hash, the canary row
        (runtime/hnsw.rs:270-330, 630-678).  key_of_node[i] = (row key columns.., field, sub index) of node i.  Nodes taken out
        with `remove` have no rows (hnsw_remove deletes them, :728-868).  After an insert / remove, codec.stored_rows_delta against
        the rows the store holds is what has to be written."""
        from .ingest import encode_index_rows
        nodes, nbrs, entry = self.export()
        degs = self.degrees()  # (not always the number of link rows: extend_candidates, rows with several vectors)
        vecs = self.export_vectors()
        gone = self.__dict__.get("_removed")
        if gone and len(nbrs) and degs is not None:
            degs = [degs[0][np.setdiff1d(np.arange(self.n), np.fromiter(gone, dtype=np.int64))]] + list(degs[1:])
        if gone and len(nbrs):
            live0 = np.setdiff1d(np.arange(self.n, dtype=np.uint32), np.fromiter(gone, dtype=np.uint32), assume_unique=False)
            nodes = [live0.astype(np.uint32)] + list(nodes[1:])
            nbrs = [nbrs[0][live0]] + list(nbrs[1:])
        level_dist = []
        for lv, (ids, tab) in enumerate(zip(nodes, nbrs)):
            if ids is None or (lv == 0 and not gone):
                ids = np.arange(tab.shape[0], dtype=np.uint32)
            d = np.zeros(tab.shape, dtype=np.float64)
            live = tab != CZ_NONE
            if live.any():
                fr = np.broadcast_to(ids[:, None], tab.shape)[live]
                pairs = np.stack([fr, tab[live]], 1).astype(np.uint32)
                d[live] = distance_batch(self.manifest.distance, vecs, vecs, pairs)
            level_dist.append(d)
        return encode_index_rows(key_of_node, vecs, nodes, nbrs, entry, DISTANCES[self.manifest.distance], level_dist,
                                 relation_id, level_degree=degs)

    def export_vectors(self) -> np.ndarray:
        out = np.empty((self.n, self.manifest.vec_dim), dtype=np.float32)
        check(_lib.lib().cz_hnsw_index_export_vectors(self._h, ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cz_hnsw_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(_lib.lib().cz_hnsw_index_bytes(self._h))

    @property
    def table_contiguous(self) -> bool:
        """cz_hnsw_index_table_contiguous: the vector table got one physically contiguous range (a placement diagnostic)"""
        return bool(_lib.lib().cz_hnsw_index_table_contiguous(self._h))

    def settle(self, ef: int = 0, trials: int = 0):
        """cz_hnsw_index_settle: placement by trial (trials = 0: only report what create / build did).
        -> (calibration ms before, after, candidates timed)"""
        a, b, t = C.c_double(0.0), C.c_double(0.0), C.c_uint32(0)
        check(_lib.lib().cz_hnsw_index_settle(self._h, int(ef), int(trials), C.byref(a), C.byref(b), C.byref(t)))
        return a.value, b.value, int(t.value)

    def hbm_probe(self, n_fetch: int = 0, reps: int = 0):
        """cz_hnsw_index_probe: (contiguous read GB/s, random whole-row fetch GB/s) over this index' vector table"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(_lib.lib().cz_hnsw_index_probe(self._h, int(n_fetch), int(reps), C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- hnsw_knn over a batch of parent tuples (host buffers) ----
    def hnsw_knn_batch(self, queries: np.ndarray, config: HnswSearch, poison: Optional[np.ndarray] = None,
                       with_n_dist: bool = False):
        """Returns (ids [B][kk], dist [B][kk] f64, count [B]) with kk = ef if a filter follows else k
        (hnsw.rs:943-947), rows ascending by distance; optionally the per-query distance-evaluation count.
        The query is converted to the index' element type first (hnsw.rs:879-884)."""
        f64 = getattr(self, "_np", np.float32) is np.float64
        q = np.ascontiguousarray(queries, dtype=np.float64 if f64 else np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.manifest.vec_dim:
            raise ValueError("query vector dimension mismatch")  # hnsw.rs:876-878
        B = q.shape[0]
        kk = config.ef if config.has_filter else config.k
        ids = np.empty((B, kk), dtype=np.uint32)
        dist = np.empty((B, kk), dtype=np.float64)
        cnt = np.empty(B, dtype=np.uint32)
        nd = np.zeros(B, dtype=np.uint64) if with_n_dist else None
        search = _lib.lib().cz_hnsw_search_batch_f64 if f64 else _lib.lib().cz_hnsw_search_batch
        check(search(self._h, ptr(q), B, kk, config.ef, int(config.radius is not None), float(config.radius or 0.0), ptr(ids), ptr(dist),
                     ptr(cnt), ptr(nd), ptr(poison), 0, None))
        return (ids, dist, cnt, nd) if with_n_dist else (ids, dist, cnt)

    # ---- filtered search with the comparisons on the device (cz_hnsw_search_filtered) ----
    def upload_column(self, values: np.ndarray):
        """one numeric value per node (f64 or i64) -> a device-resident column handle for `predicates`"""
        v = np.ascontiguousarray(values)
        if v.shape != (self.n,):
            raise ValueError("a column holds one value per node of the index")
        if v.dtype == np.int64:
            ty = _lib.CZ_COL_I64
        elif v.dtype == np.float64:
            ty = _lib.CZ_COL_F64
        else:
            raise ValueError("columns are int64 or float64")
        h = C.c_void_p()
        check(_lib.lib().cz_column_upload(ptr(v), self.n, ty, C.byref(h)))
        return DeviceColumn(h, ty)

    def hnsw_knn_batch_filtered(self, queries: np.ndarray, config: HnswSearch, predicates, with_n_dist: bool = False):
        """hnsw_knn with a filter that is a conjunction of `column OP constant` (predicates = [(DeviceColumn, op, constant)],
        op in < <= == >= > !=, constant int or float): all ef candidates are filtered on the device, k rows come back
        (hnsw.rs:943-947, 997-1006)."""
        f64 = getattr(self, "_np", np.float32) is np.float64
        q = np.ascontiguousarray(queries, dtype=np.float64 if f64 else np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.manifest.vec_dim:
            raise ValueError("query vector dimension mismatch")
        B = q.shape[0]
        arr = (_lib.Predicate * len(predicates))()
        for i, (col, op, const) in enumerate(predicates):
            is_int = isinstance(const, (int, np.integer)) and not isinstance(const, bool)
            arr[i] = _lib.Predicate(col._h, _lib.CZ_OPS[op], _lib.CZ_COL_I64 if is_int else _lib.CZ_COL_F64,
                                    0.0 if is_int else float(const), int(const) if is_int else 0)
        ids = np.empty((B, config.k), dtype=np.uint32)
        dist = np.empty((B, config.k), dtype=np.float64)
        cnt = np.empty(B, dtype=np.uint32)
        nd = np.zeros(B, dtype=np.uint64) if with_n_dist else None
        check((_lib.lib().cz_hnsw_search_filtered_f64 if f64 else _lib.lib().cz_hnsw_search_filtered)(self._h, ptr(q), B, config.k, config.ef, int(config.radius is not None),
                                                 float(config.radius or 0.0), arr, len(predicates), ptr(ids), ptr(dist),
                                                 ptr(cnt), ptr(nd), None, 0, None))
        return (ids, dist, cnt, nd) if with_n_dist else (ids, dist, cnt)

    def hnsw_knn(self, q: np.ndarray, config: HnswSearch):
        """One parent tuple: list of (node id, distance) rows, ascending (hnsw.rs:1005-1006)."""
        ids, dist, cnt = self.hnsw_knn_batch(np.asarray(q)[None, :], config)
        c = int(cnt[0])
        return list(zip(ids[0, :c].tolist(), dist[0, :c].tolist()))

    # ---- device-resident form (torch tensors on the GPU; used by bench.py and the sharded path) ----
    def hnsw_knn_batch_device(self, queries, config: HnswSearch, out_ids, out_dist, out_count, out_n_dist=None,
                              stream: int = 0):
        B = queries.shape[0]
        kk = config.ef if config.has_filter else config.k
        # (device buffers: f32 rows for an F32 index, f64 rows for an F64 one -- the caller's job, nothing is converted here)
        fn = _lib.lib().cz_hnsw_search_batch_f64 if getattr(self, "_np", np.float32) is np.float64 else _lib.lib().cz_hnsw_search_batch
        check(fn(self._h, ptr(queries), B, kk, config.ef, int(config.radius is not None), float(config.radius or 0.0), ptr(out_ids),
                 ptr(out_dist), ptr(out_count), ptr(out_n_dist), None, CZ_DEVICE_PTRS, C.c_void_p(stream)))

    def bruteforce_knn(self, queries: np.ndarray, k: int, gemm: bool = False):
        """exact k-NN by exhaustive scan; gemm=True computes the B x N dot products as one dense f32 GEMM on the
        matrix cores (Cosine / IP; every dot product is then a k-ordered fmaf chain)"""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        B = q.shape[0]
        ids = np.empty((B, k), dtype=np.uint32)
        dist = np.empty((B, k), dtype=np.float64)
        check(_lib.lib().cz_knn_bruteforce(self._h, ptr(q), B, k, ptr(ids), ptr(dist), _lib.CZ_BF_GEMM if gemm else 0, None))
        return ids, dist

    def bruteforce_knn_device(self, queries, k: int, out_ids, out_dist, stream: int = 0, gemm: bool = False):
        check(_lib.lib().cz_knn_bruteforce(self._h, ptr(queries), queries.shape[0], k, ptr(out_ids), ptr(out_dist),
                                           CZ_DEVICE_PTRS | (_lib.CZ_BF_GEMM if gemm else 0), C.c_void_p(stream)))


    def distance_batch(self, queries: np.ndarray, pairs: np.ndarray) -> np.ndarray:
        """VectorCache::dist over (query row, node) pairs against this index's resident vectors (cz_hnsw_index_distance_batch)"""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        pr = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.empty(pr.shape[0], dtype=np.float64)
        check(_lib.lib().cz_hnsw_index_distance_batch(self._h, ptr(q), q.shape[0], ptr(pr), pr.shape[0], ptr(out), 0, None))
        return out

    def distance_batch_device(self, queries, pairs, out, stream: int = 0):
        check(_lib.lib().cz_hnsw_index_distance_batch(self._h, ptr(queries), queries.shape[0], ptr(pairs), pairs.shape[0], ptr(out),
                                                      CZ_DEVICE_PTRS, C.c_void_p(stream)))


class DeviceColumn:
    """a per-node numeric column resident in HBM (cz_column)"""

    def __init__(self, handle, ty):
        self._h, self.type = handle, ty

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cz_column_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def distance_batch(distance: str, base: np.ndarray, queries: np.ndarray, pairs: np.ndarray) -> np.ndarray:
    """VectorCache::dist (hnsw.rs:66-109) == l2_dist / cos_dist / ip_dist (data/functions.rs:2185-2255) over
    (query row, base row) pairs; returns f64 like the reference."""
    base = np.ascontiguousarray(base, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    if base.shape[1] != queries.shape[1]:
        raise ValueError("requires two vectors of the same length")  # functions.rs:2190-2192
    out = np.empty(pairs.shape[0], dtype=np.float64)
    check(_lib.lib().cz_distance_batch(DISTANCES[distance], ptr(base), base.shape[0], base.shape[1], ptr(queries),
                                       queries.shape[0], ptr(pairs), pairs.shape[0], ptr(out), 0, None))
    return out


def distance_batch_f64(distance: str, base: np.ndarray, queries: np.ndarray, pairs: np.ndarray) -> np.ndarray:
    """VectorCache::dist's F64 arms (hnsw.rs:73-78, 86-95, 102-106) over (query row, base row) pairs of f64 vectors"""
    base = np.ascontiguousarray(base, dtype=np.float64)
    queries = np.ascontiguousarray(queries, dtype=np.float64)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    if base.shape[1] != queries.shape[1]:
        raise ValueError("requires two vectors of the same length")
    out = np.empty(pairs.shape[0], dtype=np.float64)
    check(_lib.lib().cz_distance_batch_f64(DISTANCES[distance], ptr(base), base.shape[0], base.shape[1], ptr(queries),
                                           queries.shape[0], ptr(pairs), pairs.shape[0], ptr(out), 0, None))
    return out


def distance_batch_device(distance: str, base, queries, pairs, out, stream: int = 0):
    check(_lib.lib().cz_distance_batch(DISTANCES[distance], ptr(base), base.shape[0], base.shape[1], ptr(queries),
                                       queries.shape[0], ptr(pairs), pairs.shape[0], ptr(out), CZ_DEVICE_PTRS,
                                       C.c_void_p(stream)))


# ---- HnswSearchRA::iter (query/ra.rs:1085-1121) + the row assembly of hnsw_knn (runtime/hnsw.rs:939-1006) ---------
@dataclass
class BaseRelation:
    """The stored relation an index hangs off: key columns then non-key columns; rows in key order."""
    keys: Sequence[str]
    non_keys: Sequence[str]
    rows: Sequence[tuple]

    def column_name(self, idx: int) -> str:
        return self.keys[idx] if idx < len(self.keys) else self.non_keys[idx - len(self.keys)]


@dataclass
class HnswSearchBinding:
    """data/program.rs:975-991: which extra columns the search binds, plus radius / filter."""
    k: int
    ef: int
    bind_field: bool = False
    bind_field_idx: bool = False
    bind_distance: bool = False
    bind_vector: bool = False
    radius: Optional[float] = None
    filter: Optional[callable] = None  # the compiled filter expression over the bound result tuple
    # the part of the filter that is a conjunction of `base column OP constant` over numeric columns: [(column index of
    # the base relation, op, constant)].  HnswSearchRA hands it to the device (cz_hnsw_search_filtered) when every value
    # of those columns is an int / a float and `filter` is None; the reference evaluates the same comparisons as filter
    # bytecode on the ef candidate rows (hnsw.rs:994-998)
    predicates: Optional[Sequence[tuple]] = None


def _value_eq(a, b) -> bool:
    """DataValue equality for a pair that is not two numbers: different kinds are different values"""
    kind = lambda x: ("null" if x is None else "bool" if isinstance(x, (bool, np.bool_)) else
                      "num" if isinstance(x, (int, float, np.integer, np.floating)) else "str" if isinstance(x, str) else
                      "bytes" if isinstance(x, (bytes, bytearray)) else "vec" if isinstance(x, np.ndarray) else
                      "list" if isinstance(x, (list, tuple)) else type(x).__name__)
    if kind(a) != kind(b):
        return False
    if kind(a) == "num":  # Num::eq is cmp == Equal (data/value.rs:531-576): an Int never equals a Float, floats by total order
        ai, bi = isinstance(a, (int, np.integer)), isinstance(b, (int, np.integer))
        if ai != bi:
            return False
        return int(a) == int(b) if ai else np.float64(a).tobytes() == np.float64(b).tobytes()
    if isinstance(a, np.ndarray):
        return a.dtype == b.dtype and a.shape == b.shape and bool(np.array_equal(a, b))
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_value_eq(x, y) for x, y in zip(a, b))
    return bool(a == b)


def _compare(a, op, b) -> bool:
    """op_lt / op_le / op_eq / op_ge / op_gt / op_neq (data/functions.rs:298-380).  Numbers: Int with Int as integers,
    Float with Float by total order (data/value.rs:595), mixed pairs as f64.  op_eq / op_neq never check types
    (:298-304, :337-343): a pair that is not two numbers is compared as DataValues (None == 5 is false, 'a' != 5 is
    true); the ordering operators call ensure_same_value_type first and fail on such a pair -- that error is kept."""
    num = lambda x: isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, np.bool_))
    if not (num(a) and num(b)):
        if op == "==":
            return _value_eq(a, b)
        if op == "!=":
            return not _value_eq(a, b)
        raise TypeError("comparison can only be done between the same datatypes")
    ai, bi = isinstance(a, (int, np.integer)), isinstance(b, (int, np.integer))
    if ai and bi:
        c = (int(a) > int(b)) - (int(a) < int(b))
    elif not ai and not bi:
        import struct
        key = lambda x: (lambda u: u ^ (0x7FFFFFFFFFFFFFFF if u < 0 else 0))(struct.unpack("<q", struct.pack("<d", float(x)))[0])
        ka, kb = key(a), key(b)
        c = (ka > kb) - (ka < kb)
    else:
        fa, fb = float(a), float(b)
        return {"<": fa < fb, "<=": fa <= fb, "==": fa == fb, ">=": fa >= fb, ">": fa > fb, "!=": fa != fb}[op]
    return {"<": c < 0, "<=": c <= 0, "==": c == 0, ">=": c >= 0, ">": c > 0, "!=": c != 0}[op]


def index_nodes(base: BaseRelation, vec_fields: Sequence[int]):
    """hnsw_put's extraction order (runtime/hnsw.rs:679-727): per row, per indexed field, a vector or every vector of
    a list -> [(row position, field, sub-index)] = the node -> CompoundKey table, and the vectors in that order."""
    nodes, vecs = [], []
    for r, t in enumerate(base.rows):
        for f in vec_fields:
            v = t[f]
            if isinstance(v, np.ndarray):
                nodes.append((r, f, -1))
                vecs.append(v)
            elif isinstance(v, (list, tuple)):
                for s, x in enumerate(v):
                    if isinstance(x, np.ndarray):
                        nodes.append((r, f, s))
                        vecs.append(x)
    return nodes, (np.stack(vecs).astype(np.float32) if vecs else np.zeros((0, 0), np.float32))


class HnswSearchRA:
    """`parent` yields tuples carrying a vector at `bind_idx`; the reference calls hnsw_knn once per parent tuple, this
    drains the parent into ONE batch, searches it with one launch and re-emits `parent ++ result` in parent order.
    `index` is anything with hnsw_knn_batch(queries, HnswSearch) -> (ids, dist, count): a GpuHnswIndex."""

    def __init__(self, index, base: BaseRelation, nodes: Sequence[tuple], search: HnswSearchBinding, bind_idx: int):
        self.index, self.base, self.nodes, self.search, self.bind_idx = index, base, list(nodes), search, bind_idx

    def _device_predicates(self, preds):
        """[(DeviceColumn, op, constant)] for cz_hnsw_search_filtered, or None when a column is not purely Int or purely
        Float over the indexed rows (Null, strings, a mix: the reference's comparison then depends on the row, or raises)"""
        cache = self.__dict__.setdefault("_columns", {})
        out = []
        for c, op, v in preds:
            if isinstance(v, bool) or not isinstance(v, (int, float, np.integer, np.floating)) or op not in _lib.CZ_OPS:
                return None
            if c not in cache:
                vals = [self.base.rows[r][c] for r, _, _ in self.nodes]
                if vals and all(isinstance(x, (int, np.integer)) and not isinstance(x, bool) for x in vals):
                    cache[c] = self.index.upload_column(np.asarray(vals, dtype=np.int64))
                elif vals and all(isinstance(x, (float, np.floating)) for x in vals):
                    cache[c] = self.index.upload_column(np.asarray(vals, dtype=np.float64))
                else:
                    cache[c] = None
            if cache[c] is None:
                return None
            out.append((cache[c], op, v))
        return out if 0 < len(out) <= 4 else None

    def iter(self, parent: Sequence[tuple]):
        sb = self.search
        qs = []
        for t in parent:
            v = t[self.bind_idx] if self.bind_idx < len(t) else None
            if not isinstance(v, np.ndarray):
                raise ValueError(f"Expected vector, got {v!r}")  # ra.rs:1106-1109
            qs.append(np.asarray(v, dtype=np.float32))
        if not qs:
            return []
        # without a filter the candidates are cut to k before rows are fetched; with one all ef survive until the
        # filter has run (hnsw.rs:943-947)
        # the radius cut (`distance > r => skip`, :952-956) is applied on the device to the rows that come back
        preds = list(sb.predicates or [])
        host_filter = sb.filter
        on_device = None
        if preds and sb.filter is None and hasattr(self.index, "hnsw_knn_batch_filtered"):
            on_device = self._device_predicates(preds)
        if preds and on_device is None:  # evaluated on the host rows below, with the reference's comparison semantics
            inner = sb.filter
            host_filter = lambda row: all(_compare(row[c], op, v) for c, op, v in preds) and (inner is None or inner(row))
        if on_device is not None:
            ids, dist, cnt = self.index.hnsw_knn_batch_filtered(np.stack(qs), HnswSearch(k=sb.k, ef=sb.ef, radius=sb.radius),
                                                                on_device)
        else:
            cfg = HnswSearch(k=sb.k, ef=sb.ef, radius=sb.radius, has_filter=host_filter is not None)
            if host_filter is None:
                cfg = HnswSearch(k=min(sb.k, sb.ef), ef=sb.ef, radius=sb.radius)
            ids, dist, cnt = self.index.hnsw_knn_batch(np.stack(qs), cfg)
        out = []
        for i, t in enumerate(parent):
            rows = []
            for j in range(int(cnt[i])):
                d = float(dist[i, j])
                if sb.radius is not None and d > sb.radius:  # :952-956
                    continue
                r, f, s = self.nodes[int(ids[i, j])]
                cand = list(self.base.rows[r])
                field_val = cand[f]
                # "make sure the order is the same as in all_bindings()" (:962): field, field_idx, distance, vector
                if sb.bind_field:
                    cand.append(self.base.column_name(f))
                if sb.bind_field_idx:
                    cand.append(None if s < 0 else s)
                if sb.bind_distance:
                    cand.append(d)
                if sb.bind_vector:
                    cand.append(field_val if s < 0 else field_val[s])
                if host_filter is not None and not host_filter(tuple(cand)):  # :994-998
                    continue
                rows.append(tuple(cand))
            for c in rows[:sb.k]:  # :1005-1006 (rows already ascending by distance)
                out.append(tuple(t) + c)
        return out

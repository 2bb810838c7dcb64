"""ctypes front-end of the CPU oracle (oracle/cozo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's `cpu_baseline` leg and by
__graft_entry__.smoke() -- never by the product package `cozo_amd`.
"parity unpinned" for everything except the two tiny reference-pinned cases (see cozo_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcozo_oracle.so")

L2, COSINE, IP = 0, 1, 2
DOT_NDARRAY, DOT_GPU, DOT_SEQ = 0, 1, 2
NONE = 0xFFFFFFFF

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cozo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcozo_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _FlatIndex(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("dim", C.c_int),
        ("metric", C.c_int),
        ("dot_mode", C.c_int),
        ("vectors", _f32p),
        ("n_levels", C.c_int),
        ("level_size", _u32p),
        ("level_width", _i32p),
        ("level_nodes", C.POINTER(_u32p)),
        ("level_nbrs", C.POINTER(_u32p)),
        ("entry", C.c_uint32),
        ("vectors64", C.POINTER(C.c_double)),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_dot_ndarray.restype = C.c_float
        L.orc_dot_ndarray.argtypes = [_f32p, _f32p, C.c_size_t]
        L.orc_dot_gpu.restype = C.c_float
        L.orc_dot_gpu.argtypes = [_f32p, _f32p, C.c_int]
        L.orc_l2_gpu.restype = C.c_float
        L.orc_l2_gpu.argtypes = [_f32p, _f32p, C.c_int]
        L.orc_distance.restype = C.c_double
        L.orc_distance.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int]
        L.orc_distance_pairs.restype = None
        L.orc_distance_pairs.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int, _u32p, C.c_uint64, _f64p]
        L.orc_hnsw_new.restype = C.c_void_p
        L.orc_hnsw_new.argtypes = [C.c_int] * 7
        L.orc_hnsw_free.argtypes = [C.c_void_p]
        L.orc_hnsw_insert.restype = C.c_int
        L.orc_hnsw_insert.argtypes = [C.c_void_p, _f32p, C.c_uint32, _i32p]
        L.orc_hnsw_size.restype = C.c_uint32
        L.orc_hnsw_size.argtypes = [C.c_void_p]
        L.orc_hnsw_n_levels.restype = C.c_int
        L.orc_hnsw_n_levels.argtypes = [C.c_void_p]
        L.orc_hnsw_entry.restype = C.c_uint32
        L.orc_hnsw_entry.argtypes = [C.c_void_p]
        L.orc_hnsw_level_size.restype = C.c_uint32
        L.orc_hnsw_level_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_hnsw_level_width.restype = C.c_int
        L.orc_hnsw_level_width.argtypes = [C.c_void_p, C.c_int]
        L.orc_hnsw_export_level.argtypes = [C.c_void_p, C.c_int, _u32p, _u32p]
        L.orc_hnsw_dist_count.restype = C.c_uint64
        L.orc_hnsw_dist_count.argtypes = [C.c_void_p]
        L.orc_hnsw_link_rows.restype = C.c_uint64
        L.orc_hnsw_link_rows.argtypes = [C.c_void_p, C.c_int]
        L.orc_hnsw_set_row_of.restype = None
        L.orc_hnsw_set_row_of.argtypes = [C.c_void_p, _u32p, C.c_uint32]
        L.orc_hnsw_set_key_order.restype = None
        L.orc_hnsw_set_key_order.argtypes = [C.c_void_p, _u32p, C.c_uint32]
        L.orc_hnsw_remove.restype = C.c_int
        L.orc_hnsw_remove.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_hnsw_dangling_links.restype = C.c_uint64
        L.orc_hnsw_dangling_links.argtypes = [C.c_void_p]
        L.orc_hnsw_degree.restype = C.c_double
        L.orc_hnsw_degree.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.orc_hnsw_knn.restype = C.c_int
        L.orc_hnsw_knn.argtypes = [C.POINTER(_FlatIndex), _f32p, C.c_int, C.c_int, C.c_int, C.c_double, _u32p, _f64p,
                                   _u64p]
        L.orc_hnsw_knn_batch.restype = None
        L.orc_hnsw_knn_batch.argtypes = [C.POINTER(_FlatIndex), _f32p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                         C.c_double, _u32p, _f64p, _u32p, _u64p, C.c_int]
        L.orc_bruteforce_knn.restype = None
        L.orc_bruteforce_knn.argtypes = [C.c_int, C.c_int, _f32p, C.c_uint32, C.c_int, _f32p, C.c_uint32, C.c_int,
                                         _u32p, _f64p, C.c_int]
        L.orc_assign_ids.restype = C.c_uint32
        L.orc_assign_ids.argtypes = [_i64p, _i64p, C.c_uint64, _u32p, _u32p, _i64p]
        L.orc_build_csr.restype = None
        L.orc_build_csr.argtypes = [C.c_uint32, C.c_uint64, _u32p, _u32p, _f32p, C.c_int, _u64p, _u32p, _f32p]
        L.orc_pagerank_inplace_lockstep.restype = C.c_int
        L.orc_pagerank_inplace_lockstep.argtypes = [C.c_uint32, _u64p, _u32p, _u32p, C.c_float, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    _f32p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.orc_pagerank_mode.restype = C.c_int
        L.orc_pagerank_mode.argtypes = [C.c_uint32, _u64p, _u32p, _u32p, C.c_float, C.c_double, C.c_uint32, C.c_int, C.c_int,
                                        _f32p, _u32p, C.POINTER(C.c_double)]
        L.orc_pagerank.restype = C.c_int
        L.orc_pagerank.argtypes = [C.c_uint32, _u64p, _u32p, _u32p, C.c_float, C.c_double, C.c_uint32, _f32p, _u32p,
                                   _f64p, C.c_int]
        L.orc_shortest_path_bfs.restype = None
        L.orc_shortest_path_bfs.argtypes = [C.c_uint32, _u64p, _u32p, C.c_uint32, _u32p, C.c_uint32, _u32p]
        L.orc_bfs_order.restype = C.c_uint32
        L.orc_bfs_order.argtypes = [C.c_uint32, _u64p, _u32p, C.c_uint32, _u8p, _u32p, _u32p]
        L.orc_tarjan_groups.restype = C.c_uint32
        L.orc_tarjan_groups.argtypes = [C.c_uint32, _u64p, _u32p, _u32p]
        L.orc_clustering_coefficients.restype = None
        L.orc_clustering_coefficients.argtypes = [C.c_uint32, _u64p, _u32p, C.POINTER(C.c_double), _u64p, _u32p]
        L.orc_dijkstra.restype = None
        L.orc_dijkstra.argtypes = [C.c_uint32, _u64p, _u32p, _f32p, C.c_uint32, _u32p, C.c_uint32, _f32p, _u32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


# ------------------------------------------------------------------ distances
def dot_ndarray(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot_ndarray(_p(a, _f32p), _p(b, _f32p), a.size))


def distance(metric: int, a, b, dot_mode: int = DOT_NDARRAY) -> float:
    a, b = _f32(a), _f32(b)
    assert a.size == b.size
    return float(lib().orc_distance(metric, dot_mode, _p(a, _f32p), _p(b, _f32p), a.size))


def distance_pairs_f64(metric: int, base, queries, pairs, dot_mode: int = DOT_NDARRAY) -> np.ndarray:
    """VectorCache::dist's F64 arms over (query row, base row) pairs"""
    base = np.ascontiguousarray(base, dtype=np.float64)
    queries = np.ascontiguousarray(queries, dtype=np.float64)
    pairs = _u32(pairs)
    out = np.empty(pairs.shape[0], dtype=np.float64)
    f64p = C.POINTER(C.c_double)
    fn = lib().orc_distance_pairs_f64
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, f64p, f64p, C.c_int, _u32p, C.c_uint64, f64p]
    fn(metric, dot_mode, base.ctypes.data_as(f64p), queries.ctypes.data_as(f64p), base.shape[1], _p(pairs, _u32p), pairs.shape[0],
       out.ctypes.data_as(f64p))
    return out


def distance_pairs(metric: int, base, queries, pairs, dot_mode: int = DOT_NDARRAY) -> np.ndarray:
    base, queries, pairs = _f32(base), _f32(queries), _u32(pairs)
    out = np.empty(pairs.shape[0], dtype=np.float64)
    lib().orc_distance_pairs(metric, dot_mode, _p(base, _f32p), _p(queries, _f32p), base.shape[1], _p(pairs, _u32p),
                             pairs.shape[0], _p(out, _f64p))
    return out


# ------------------------------------------------------------------ HNSW
def random_levels(n: int, m: int, seed: int) -> np.ndarray:
    """`get_random_level` (runtime/hnsw.rs:46-52) with an injectable generator:
    level = floor(-ln(U) * 1/ln(m)), returned as the non-negative -layer."""
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    u = np.where(u == 0.0, np.finfo(np.float64).tiny, u)
    return np.floor(-np.log(u) * (1.0 / np.log(m))).astype(np.int32)


class FlatIndex:
    """The flat HNSW layout handed to both the oracle search and libcozo_gpu (include/cozo_gpu.h)."""

    def __init__(self, vectors, metric, level_nodes, level_nbrs, entry, f64=False):
        """f64: an index of f64 vectors (VecElementType::F64): distances by VectorCache::dist's F64 arms, queries are f64 rows"""
        self.f64 = bool(f64)
        self.vectors = np.ascontiguousarray(vectors, dtype=np.float64) if self.f64 else _f32(vectors)
        self.n, self.dim = self.vectors.shape
        self.metric = metric
        self.level_nodes = [_u32(x) for x in level_nodes]
        self.level_nbrs = [_u32(x) for x in level_nbrs]
        self.n_levels = len(self.level_nbrs)
        self.entry = int(entry)
        self.level_size = np.array([x.shape[0] for x in self.level_nbrs], dtype=np.uint32)
        self.level_width = np.array([x.shape[1] for x in self.level_nbrs], dtype=np.int32)

    def _cstruct(self, dot_mode):
        nodes = (_u32p * max(self.n_levels, 1))(*[_p(x, _u32p) for x in self.level_nodes])
        nbrs = (_u32p * max(self.n_levels, 1))(*[_p(x, _u32p) for x in self.level_nbrs])
        f64p = C.POINTER(C.c_double)
        s = _FlatIndex(self.n, self.dim, self.metric, dot_mode, None if self.f64 else _p(self.vectors, _f32p), self.n_levels,
                       _p(self.level_size, _u32p), _p(self.level_width, _i32p), nodes, nbrs, self.entry,
                       self.vectors.ctypes.data_as(f64p) if self.f64 else f64p())
        s._keep = (nodes, nbrs)
        return s

    def knn_batch(self, queries, k, ef, radius=None, dot_mode=DOT_NDARRAY, threads=1):
        # (the query is converted to the index' element type before the search, runtime/hnsw.rs:879-884)
        queries = np.ascontiguousarray(queries, dtype=np.float64) if self.f64 else _f32(queries)
        B = queries.shape[0]
        ids = np.empty((B, k), dtype=np.uint32)
        dist = np.empty((B, k), dtype=np.float64)
        cnt = np.empty(B, dtype=np.uint32)
        nd = C.c_uint64(0)
        s = self._cstruct(dot_mode)
        lib().orc_hnsw_knn_batch(C.byref(s), queries.ctypes.data_as(_f32p), B, k, ef, int(radius is not None),
                                 float(radius or 0.0), _p(ids, _u32p), _p(dist, _f64p), _p(cnt, _u32p), C.byref(nd),
                                 threads)
        return ids, dist, cnt, nd.value


class HnswBuilder:
    """Sequential index construction exactly as `hnsw_put` over rows in key order."""

    def __init__(self, dim, metric, m, ef_construction, extend_candidates=False, keep_pruned_connections=False,
                 dot_mode=DOT_NDARRAY):
        self.dim, self.metric, self.m = dim, metric, m
        self._h = lib().orc_hnsw_new(dim, metric, m, ef_construction, int(extend_candidates),
                                     int(keep_pruned_connections), dot_mode)
        self._vecs = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_hnsw_free(self._h)
            self._h = None

    def insert(self, vectors, levels):
        vectors = _f32(vectors)
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        assert vectors.shape == (levels.size, self.dim)
        rc = lib().orc_hnsw_insert(self._h, _p(vectors, _f32p), vectors.shape[0], _p(levels, _i32p))
        assert rc == 0
        self._vecs.append(vectors)

    @property
    def size(self):
        return lib().orc_hnsw_size(self._h)

    def dist_count(self):
        return lib().orc_hnsw_dist_count(self._h)

    def link_rows(self, include_ignored=False):
        return lib().orc_hnsw_link_rows(self._h, int(include_ignored))

    def set_key_order(self, key_rank):
        """key_rank[node] = position of the node's key among all keys (held and to come); None: ids are key order"""
        if key_rank is None:
            lib().orc_hnsw_set_key_order(self._h, None, 0)
        else:
            r = _u32(key_rank)
            lib().orc_hnsw_set_key_order(self._h, _p(r, _u32p), r.size)

    def set_row_of(self, row_of):
        """row_of[node] = the base row of the node's vector (nodes held and to come): links inside one row are invisible to
        every reader (hnsw_get_neighbours, hnsw.rs:609-610)"""
        if row_of is None:
            lib().orc_hnsw_set_row_of(self._h, None, 0)
        else:
            r = _u32(row_of)
            lib().orc_hnsw_set_row_of(self._h, _p(r, _u32p), r.size)

    def remove(self, nodes):
        """hnsw_remove_vec (hnsw.rs:754-868) for every listed node, in order; how many were indexed"""
        return sum(lib().orc_hnsw_remove(self._h, int(v)) for v in nodes)

    def dangling_links(self):
        """rows the reference leaves pointing at removed nodes (a search following one fails in ensure_key, hnsw.rs:133)"""
        return lib().orc_hnsw_dangling_links(self._h)

    def degree(self, node, level):
        return lib().orc_hnsw_degree(self._h, int(node), int(level))

    def export(self) -> FlatIndex:
        L = lib()
        nl = L.orc_hnsw_n_levels(self._h)
        nodes, nbrs = [], []
        for lv in range(nl):
            sz, w = L.orc_hnsw_level_size(self._h, lv), L.orc_hnsw_level_width(self._h, lv)
            ids = np.empty(sz, dtype=np.uint32)
            tab = np.empty((sz, w), dtype=np.uint32)
            L.orc_hnsw_export_level(self._h, lv, _p(ids, _u32p), _p(tab, _u32p))
            nodes.append(ids)
            nbrs.append(tab)
        vecs = np.concatenate(self._vecs, axis=0) if self._vecs else np.zeros((0, self.dim), np.float32)
        return FlatIndex(vecs, self.metric, nodes, nbrs, L.orc_hnsw_entry(self._h) if nl else NONE)


def bruteforce_knn(metric, base, queries, k, dot_mode=DOT_NDARRAY, threads=8):
    base, queries = _f32(base), _f32(queries)
    B = queries.shape[0]
    ids = np.empty((B, k), dtype=np.uint32)
    dist = np.empty((B, k), dtype=np.float64)
    lib().orc_bruteforce_knn(metric, dot_mode, _p(base, _f32p), base.shape[0], base.shape[1], _p(queries, _f32p), B, k,
                             _p(ids, _u32p), _p(dist, _f64p), threads)
    return ids, dist


# ------------------------------------------------------------------ graphs
def assign_ids(frm, to):
    """first-appearance dense ids over rows in scan order (fixed_rule/mod.rs:144-186)."""
    frm = np.ascontiguousarray(frm, dtype=np.int64)
    to = np.ascontiguousarray(to, dtype=np.int64)
    E = frm.size
    fi = np.empty(E, dtype=np.uint32)
    ti = np.empty(E, dtype=np.uint32)
    ind = np.empty(2 * E + 1, dtype=np.int64)
    n = lib().orc_assign_ids(_p(frm, _i64p), _p(to, _i64p), E, _p(fi, _u32p), _p(ti, _u32p), _p(ind, _i64p))
    return fi, ti, ind[:n].copy()


def build_csr(n, src, dst, weights=None, undirected=False):
    src, dst = _u32(src), _u32(dst)
    E = src.size
    Et = 2 * E if undirected else E
    off = np.empty(n + 1, dtype=np.uint64)
    tgt = np.empty(Et, dtype=np.uint32)
    w = _f32(weights) if weights is not None else None
    wo = np.empty(Et, dtype=np.float32) if weights is not None else None
    lib().orc_build_csr(n, E, _p(src, _u32p), _p(dst, _u32p), _p(w, _f32p), int(undirected), _p(off, _u64p),
                        _p(tgt, _u32p), _p(wo, _f32p))
    return (off, tgt, wo) if weights is not None else (off, tgt)


def pagerank(n, in_off, in_src, out_deg, damping=0.85, tolerance=1e-4, max_iter=10, threads=1):
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_src, out_deg = _u32(in_src), _u32(out_deg)
    scores = np.empty(n, dtype=np.float32)
    it = C.c_uint32(0)
    err = C.c_double(0)
    lib().orc_pagerank(n, _p(in_off, _u64p), _p(in_src, _u32p), _p(out_deg, _u32p), np.float32(damping),
                       float(tolerance), max_iter, _p(scores, _f32p), C.byref(it), C.byref(err), threads)
    return scores, it.value, err.value


PR_JACOBI, PR_INPLACE = 0, 1


def pagerank_mode(n, in_off, in_src, out_deg, damping=0.85, tolerance=1e-4, max_iter=10, mode=PR_JACOBI, err_f64_diff=False):
    """both readings of graph 0.3.1's loop on ONE thread (cozo_oracle.c: orc_pagerank_mode)"""
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_src, out_deg = _u32(in_src), _u32(out_deg)
    scores = np.empty(n, dtype=np.float32)
    it = C.c_uint32(0)
    err = C.c_double(0)
    lib().orc_pagerank_mode(n, _p(in_off, _u64p), _p(in_src, _u32p), _p(out_deg, _u32p), np.float32(damping),
                            float(tolerance), max_iter, int(mode), int(bool(err_f64_diff)), _p(scores, _f32p), C.byref(it),
                            C.byref(err))
    return scores, it.value, err.value


def pagerank_inplace_lockstep(n, in_off, in_src, out_deg, damping=0.85, tolerance=1e-4, max_iter=10, threads=8, chunk=16384):
    """the in-place reading on `threads` rayon threads under one deterministic lockstep schedule of the crate's chunks
    (cozo_oracle.c: orc_pagerank_inplace_lockstep); threads=1 is pagerank_mode(mode=PR_INPLACE)"""
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_src, out_deg = _u32(in_src), _u32(out_deg)
    scores = np.empty(n, dtype=np.float32)
    it = C.c_uint32(0)
    err = C.c_double(0)
    lib().orc_pagerank_inplace_lockstep(n, _p(in_off, _u64p), _p(in_src, _u32p), _p(out_deg, _u32p), np.float32(damping),
                                        float(tolerance), max_iter, int(threads), int(chunk), _p(scores, _f32p), C.byref(it), C.byref(err))
    return scores, it.value, err.value


def shortest_path_bfs(n, off, tgt, start, goals):
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt, goals = _u32(tgt), _u32(goals)
    parent = np.empty(n, dtype=np.uint32)
    lib().orc_shortest_path_bfs(n, _p(off, _u64p), _p(tgt, _u32p), start, _p(goals, _u32p), goals.size,
                                _p(parent, _u32p))
    return parent


def bfs_order(n, off, tgt, start, visited=None, parent=None):
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt = _u32(tgt)
    if visited is None:
        visited = np.zeros(n, dtype=np.uint8)
    if parent is None:
        parent = np.full(n, NONE, dtype=np.uint32)
    order = np.empty(n, dtype=np.uint32)
    c = lib().orc_bfs_order(n, _p(off, _u64p), _p(tgt, _u32p), start, _p(visited, _u8p), _p(parent, _u32p),
                            _p(order, _u32p))
    return order[:c].copy(), parent, visited


def path_from_parent(parent, start, goal):
    """route reconstruction of shortest_path_bfs.rs:85-99; None when no backtrace entry."""
    if parent[goal] == NONE:
        return None
    route, cur = [], int(goal)
    while cur != start:
        route.append(cur)
        cur = int(parent[cur])
    route.append(int(start))
    return route[::-1]


def tarjan_groups(n, off, tgt):
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt = _u32(tgt)
    grp = np.empty(n, dtype=np.uint32)
    k = lib().orc_tarjan_groups(n, _p(off, _u64p), _p(tgt, _u32p), _p(grp, _u32p))
    return grp, k


def clustering_coefficients(n, off, tgt):
    """algos/triangles.rs:70-110 on the symmetrised out-CSR -> (cc f64 [n], n_triangles u64 [n], degree u32 [n])"""
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt = _u32(tgt)
    cc = np.empty(n, dtype=np.float64)
    tri = np.empty(n, dtype=np.uint64)
    deg = np.empty(n, dtype=np.uint32)
    lib().orc_clustering_coefficients(n, _p(off, _u64p), _p(tgt, _u32p), cc.ctypes.data_as(C.POINTER(C.c_double)),
                                      _p(tri, _u64p), _p(deg, _u32p))
    return cc, tri, deg


def clustering_coefficients_sample(n, off, tgt, first=0, step=16, max_seconds=15.0):
    """triangles.rs:70-110 on the nodes first, first + step, ... until max_seconds are spent
    -> (node ids processed, their n_triangles, adjacency entries of their rows)"""
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt = _u32(tgt)
    tri = np.zeros(n, dtype=np.uint64)
    ne = C.c_uint64(0)
    fn = lib().orc_clustering_coefficients_sample
    fn.restype = C.c_uint64
    done = int(fn(C.c_uint32(n), _p(off, _u64p), _p(tgt, _u32p), C.c_uint32(first), C.c_uint32(step), C.c_double(max_seconds), _p(tri, _u64p),
                  C.byref(ne)))
    nodes = (first + step * np.arange(done, dtype=np.int64)).astype(np.int64)
    return nodes, tri[nodes], int(ne.value)


def dijkstra(n, off, tgt, w, start, goals=None):
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt, w = _u32(tgt), _f32(w)
    g = _u32(goals) if goals is not None else None
    dist = np.empty(n, dtype=np.float32)
    parent = np.empty(n, dtype=np.uint32)
    lib().orc_dijkstra(n, _p(off, _u64p), _p(tgt, _u32p), _p(w, _f32p), start, _p(g, _u32p),
                       0 if g is None else g.size, _p(dist, _f32p), _p(parent, _u32p))
    return dist, parent


def betweenness(n, off, tgt, w, max_paths=10_000_000):
    """BetweennessCentrality by literal enumeration of all shortest paths (small graphs) -> f32 [n]"""
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt, w = _u32(tgt), _f32(w)
    out = np.empty(n, dtype=np.float32)
    fn = lib().orc_betweenness
    fn.restype = C.c_int
    rc = fn(n, _p(off, _u64p), _p(tgt, _u32p), _p(w, _f32p), _p(out, _f32p), C.c_uint64(max_paths))
    if rc != 0:
        raise RuntimeError("too many shortest paths to enumerate")
    return out


def lp_colouring(n, off, tgt):
    """the colour classes the GPU LabelPropagation rule processes in order (orc_lp_colouring) -> (colour u32 [n], n_colours)"""
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt = _u32(tgt)
    colour = np.empty(n, dtype=np.uint32)
    fn = lib().orc_lp_colouring
    fn.restype = C.c_uint32
    k = fn(n, _p(off, _u64p), _p(tgt, _u32p), _p(colour, _u32p))
    return colour, int(k)


def label_propagation_in_order(n, off, tgt, w, order, max_iter=10):
    """label_propagation.rs:56-109 with the node order of every iteration and the tie-break (smallest label) handed in
    -> (labels u32 [n], iterations)"""
    off = np.ascontiguousarray(off, dtype=np.uint64)
    tgt, w, order = _u32(tgt), _f32(w), _u32(order)
    labels = np.empty(n, dtype=np.uint32)
    fn = lib().orc_label_propagation_in_order
    fn.restype = C.c_int
    it = fn(n, _p(off, _u64p), _p(tgt, _u32p), _p(w, _f32p), _p(order, _u32p), C.c_uint32(max_iter), _p(labels, _u32p))
    if it < 0:
        raise RuntimeError("a best score is NaN (the reference panics)")
    return labels, int(it)


def label_propagation(n, off, tgt, w, max_iter=10):
    """the execution the GPU rule fixes: colour classes in ascending order, ids ascending inside a class"""
    colour, _ = lp_colouring(n, off, tgt)
    order = np.lexsort((np.arange(n), colour)).astype(np.uint32)
    return label_propagation_in_order(n, off, tgt, w, order, max_iter)

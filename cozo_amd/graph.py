"""Thin array-level wrappers over the graph entry points of libcozo_gpu (include/cozo_gpu.h).

Graphs are the CSR that `FixedRuleInputRelation::as_directed_graph` builds (fixed_rule/mod.rs:136-200).
The rule-level mirror (options, payload, temp store) lives in cozo_amd/fixed_rule.py.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import CZ_NONE, check, ptr


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _csr32(off, tgt):
    off = np.asarray(off)
    if off.size and int(off[-1]) >= 0xFFFFFFFF:
        raise _lib.CozoGpuError(_lib.CZ_E_UNSUPPORTED, "E must be < 2^32-1")
    return _u32(off), _u32(tgt)


_PR_MODES = {None: 0, "gather": _lib.CZ_PR_GATHER, "blocked": _lib.CZ_PR_BLOCKED, "accumulate": _lib.CZ_PR_ACCUMULATE}


def pagerank(in_off, in_src, out_deg, damping=0.85, tolerance=1e-4, max_iter=10, poison=None, cache_key=None,
             mode: Optional[str] = None, timing: Optional[dict] = None):
    """graph::page_rank as called by PageRank::run (pagerank.rs:47-50): (scores f32[N], iterations, error).
    cache_key = (hi, lo): the caller's identity of (relation, snapshot); the device layout is then kept between calls
    (cz_pagerank_cached).  timing: filled with where the time went."""
    in_off, in_src = _csr32(in_off, in_src)
    out_deg = _u32(out_deg)
    N = out_deg.size
    scores = np.empty(N, dtype=np.float32)
    it = C.c_uint32(0)
    err = C.c_double(0.0)
    tm = _lib.PagerankTiming()
    hi, lo = cache_key if cache_key is not None else (0, 0)
    flags = _PR_MODES[mode]
    check(_lib.lib().cz_pagerank_cached(int(hi), int(lo), ptr(in_off), ptr(in_src), ptr(out_deg), N, in_src.size,
                                        np.float32(damping), float(tolerance), int(max_iter), flags, ptr(scores),
                                        C.byref(it), C.byref(err), ptr(poison), C.byref(tm)))
    if timing is not None:
        timing.update(h2d_ms=tm.h2d_ms, plan_build_ms=tm.plan_build_ms, iterate_ms=tm.iterate_ms, d2h_ms=tm.d2h_ms,
                      cache_hit=bool(tm.cache_hit))
    return scores, it.value, err.value


def pagerank_inplace(in_off, in_src, out_deg, damping=0.85, tolerance=1e-4, max_iter=10, err_f64_diff=False, poison=None):
    """cz_pagerank_inplace: graph::page_rank under the reading that refreshes a node's contribution INSIDE the sweep (the reference's
    one-thread execution, an ascending Gauss-Seidel sweep) -> (scores f32[N], iterations, error, launches per sweep)"""
    in_off, in_src = _csr32(in_off, in_src)
    out_deg = _u32(out_deg)
    N = out_deg.size
    scores = np.empty(N, dtype=np.float32)
    it, lv = C.c_uint32(0), C.c_uint32(0)
    err = C.c_double(0.0)
    check(_lib.lib().cz_pagerank_inplace(ptr(in_off), ptr(in_src), ptr(out_deg), N, in_src.size, np.float32(damping), float(tolerance),
                                         int(max_iter), _lib.CZ_PR_ERR_F64_DIFF if err_f64_diff else 0, ptr(scores), C.byref(it),
                                         C.byref(err), C.byref(lv), ptr(poison)))
    return scores, it.value, err.value, lv.value


class InplacePageRankPlan:
    """Resident PageRank under the in-place reading of graph::page_rank (cz_pagerank_inplace_plan_*, csrc/pagerank_inplace.hip):
    the level-scheduled ascending Gauss-Seidel sweep with its static layout kept in HBM."""

    def __init__(self, in_off, in_src, out_deg, damping=0.85, device_ptrs=False, err_f64_diff=False, as_jacobi=False):
        """host arrays, or (device_ptrs=True) uint32 device tensors (offsets [N+1], sources [E], out-degrees [N])"""
        if not device_ptrs:
            in_off, in_src = _csr32(in_off, in_src)
            out_deg = _u32(out_deg)
            N, E = out_deg.size, in_src.size
        else:
            N, E = int(out_deg.numel()), int(in_src.numel())
        h = C.c_void_p()
        check(_lib.lib().cz_pagerank_inplace_plan_create(ptr(in_off), ptr(in_src), ptr(out_deg), N, E, np.float32(damping),
                                                         (_lib.CZ_DEVICE_PTRS if device_ptrs else 0)
                                                         | (_lib.CZ_PR_ERR_F64_DIFF if err_f64_diff else 0)
                                                         | (_lib.CZ_PR_INPLACE_AS_JACOBI if as_jacobi else 0), C.byref(h)))
        self._h, self.N, self.E = h, N, E

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cz_pagerank_inplace_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, tolerance=1e-4, max_iter=10, poison=None, stream: int = 0):
        """graph::page_rank's loop from the initial state -> (iterations, error)"""
        it, err = C.c_uint32(0), C.c_double(0.0)
        check(_lib.lib().cz_pagerank_inplace_plan_run(self._h, float(tolerance), int(max_iter), C.byref(it), C.byref(err), ptr(poison),
                                                      C.c_void_p(stream)))
        return it.value, err.value

    def init(self, stream: int = 0):
        check(_lib.lib().cz_pagerank_inplace_plan_init(self._h, C.c_void_p(stream)))

    def sweeps(self, n, stream: int = 0):
        """n more sweeps on `stream`, nothing read back"""
        check(_lib.lib().cz_pagerank_inplace_plan_sweeps(self._h, int(n), C.c_void_p(stream)))

    def read_scores(self, out=None, stream: int = 0):
        """scores [N] in the caller's numbering: into a device tensor, or returned as a numpy array"""
        if out is not None:
            check(_lib.lib().cz_pagerank_inplace_plan_read_scores(self._h, ptr(out), _lib.CZ_DEVICE_PTRS, C.c_void_p(stream)))
            return out
        host = np.empty(self.N, dtype=np.float32)
        check(_lib.lib().cz_pagerank_inplace_plan_read_scores(self._h, ptr(host), 0, C.c_void_p(stream)))
        return host

    @property
    def info(self) -> dict:
        a = np.zeros(16, dtype=np.uint64)
        b, h = C.c_double(0), C.c_double(0)
        check(_lib.lib().cz_pagerank_inplace_plan_info(self._h, ptr(a), C.byref(b), C.byref(h)))
        d = dict(zip(("levels", "row_blocks", "expand_items", "long_rows", "urgent_gap", "slice_width", "launches_per_sweep", "graph_replay",
                      "x_edges", "y_edges", "urgent_edges", "long_row_edges", "x_positions", "y_positions"), (int(x) for x in a)))
        d["host_build_ms"], d["upload_ms"] = b.value, h.value
        return d


class PageRankPlan:
    """Resident / row-sharded PageRank (cz_pagerank_plan_*): rows [row_begin,row_end) of the in-CSR."""

    def __init__(self, in_off_local, in_src, out_deg, N, row_begin, row_end, damping=0.85, device_ptrs=False,
                 mode: Optional[str] = None):
        """host arrays, or (device_ptrs=True) uint32 device tensors already resident in HBM.
        mode: None (chosen from the shard's shape) | "gather" | "blocked" | "accumulate" -- the device formulations of
        the sweep (csrc/pagerank.hip); each gives the reference's scores bit for bit."""
        if not device_ptrs:
            in_off_local, in_src = _csr32(in_off_local, in_src)
            out_deg = _u32(out_deg)
        h = C.c_void_p()
        check(_lib.lib().cz_pagerank_plan_create(ptr(in_off_local), ptr(in_src), ptr(out_deg), N, row_begin, row_end,
                                                 np.float32(damping), C.byref(h),
                                                 (_lib.CZ_DEVICE_PTRS if device_ptrs else 0)
                                                 | _PR_MODES[mode]))
        self._h = h
        self.N, self.row_begin, self.row_end = N, row_begin, row_end

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cz_pagerank_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def blocked(self) -> bool:
        return bool(_lib.lib().cz_pagerank_plan_is_blocked(self._h))

    @property
    def formulation(self) -> str:
        return {1: "gather", 2: "blocked", 3: "accumulate"}[int(_lib.lib().cz_pagerank_plan_formulation(self._h))]

    @property
    def shape(self) -> dict:
        """slices / groups / workgroups of the plan (measurement scripts)"""
        a = np.zeros(12, dtype=np.uint32)
        check(_lib.lib().cz_pagerank_plan_shape(self._h, ptr(a)))
        return dict(zip(("slices", "slice_width", "groups", "waves", "rows_per_group", "acc_workgroups", "tile_blocks", "hub_rows",
                         "pieces", "group_edges", "stream_positions"), (int(x) for x in a)))

    @property
    def timing(self):
        """(h2d_ms, build_ms): what creating the plan cost"""
        a, b = C.c_double(0), C.c_double(0)
        check(_lib.lib().cz_pagerank_plan_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def edges(self) -> int:
        return int(_lib.lib().cz_pagerank_plan_edges(self._h))

    def init(self, contrib, stream: int = 0):
        check(_lib.lib().cz_pagerank_plan_init(self._h, ptr(contrib), C.c_void_p(stream)))

    def step(self, contrib_in, contrib_out, err_out, stream: int = 0):
        check(_lib.lib().cz_pagerank_plan_step(self._h, ptr(contrib_in), ptr(contrib_out), ptr(err_out),
                                               C.c_void_p(stream)))

    def scores_ptr(self) -> int:
        return int(_lib.lib().cz_pagerank_plan_scores(self._h) or 0)

    def read_scores(self, out=None, stream: int = 0):
        """this shard's scores [row_end-row_begin]: into a device tensor, or returned as a numpy array."""
        if out is not None:
            check(_lib.lib().cz_pagerank_plan_read_scores(self._h, ptr(out), _lib.CZ_DEVICE_PTRS, C.c_void_p(stream)))
            return out
        host = np.empty(self.row_end - self.row_begin, dtype=np.float32)
        check(_lib.lib().cz_pagerank_plan_read_scores(self._h, ptr(host), 0, C.c_void_p(stream)))
        return host


class DeviceGraph:
    """a relation's CSR resident on the device (cz_graph_upload, or cz_graph_acquire / cz_graph_release under the caller's
    (relation id, snapshot) key): bfs / connected_components / sssp take one in place of the host arrays and skip the upload.
    `with DeviceGraph.acquire(key, off, tgt, w) as g:` gives the graph back to the library's cache on exit."""

    def __init__(self, off, tgt, weights=None, _handle=None, _key=None, hit=False):
        self.key, self.cache_hit = _key, hit
        if _handle is not None:
            self._h = _handle
            return
        off, tgt = _csr32(off, tgt)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        h = C.c_void_p()
        check(_lib.lib().cz_graph_upload(ptr(off), ptr(tgt), ptr(w), off.size - 1, tgt.size, C.byref(h)))
        self._h = h
        self.n = off.size - 1

    @classmethod
    def acquire(cls, key, off, tgt, weights=None):
        off, tgt = _csr32(off, tgt)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        h, hit = C.c_void_p(), C.c_int(0)
        check(_lib.lib().cz_graph_acquire(int(key[0]), int(key[1]), ptr(off), ptr(tgt), ptr(w), off.size - 1, tgt.size, C.byref(h),
                                          C.byref(hit)))
        g = cls(None, None, _handle=h, _key=(int(key[0]), int(key[1])), hit=bool(hit.value))
        g.n = off.size - 1
        return g

    def close(self):
        if getattr(self, "_h", None):
            if self.key is not None:
                _lib.lib().cz_graph_release(self.key[0], self.key[1], self._h)
            else:
                _lib.lib().cz_graph_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def bfs_shared(out_off, out_tgt, starts, poison=None, on_level=None):
    """cz_bfs_shared: Bfs::run's traversal (`visited` / `backtrace` shared by the starts) with O(N) outputs:
    (parent [N], order [N], first [n_starts + 1]) -- start i discovered order[first[i]:first[i + 1]].
    on_level(start, nodes) (cz_bfs_shared_until): called after every level with the nodes it discovered; a true return stops the
    traversal there (no further level, no further start) -- the rule's `limit` (algos/bfs.rs:88-91).  What it raises comes out of
    this call."""
    out_off, out_tgt = _csr32(out_off, out_tgt)
    N = out_off.size - 1
    starts = _u32(starts)
    parent = np.empty(N, dtype=np.uint32)
    order = np.empty(N, dtype=np.uint32)
    first = np.zeros(starts.size + 1, dtype=np.uint32)
    if on_level is None:
        check(_lib.lib().cz_bfs_shared(ptr(out_off), ptr(out_tgt), N, out_tgt.size, ptr(starts), starts.size, ptr(parent), ptr(order),
                                       ptr(first), ptr(poison)))
        return parent, order, first
    raised = []

    def level(_ctx, start, nodes, n):
        try:
            return 1 if on_level(int(start), np.ctypeslib.as_array(nodes, shape=(n,))) else 0
        except BaseException as e:  # noqa: BLE001  (it must not unwind through the C frames)
            raised.append(e)
            return 1

    cb = _lib.BFS_LEVEL_FN(level)
    check(_lib.lib().cz_bfs_shared_until(ptr(out_off), ptr(out_tgt), N, out_tgt.size, ptr(starts), starts.size, cb, None, ptr(parent),
                                         ptr(order), ptr(first), ptr(poison)))
    if raised:
        raise raised[0]
    return parent, order, first


def bfs(out_off, out_tgt, starts, goals=None, share_visited=False, want_depth=False, want_order=False, poison=None, out=None):
    """cz_bfs; `out_off` may be a DeviceGraph (then `out_tgt` is ignored): cz_bfs_on.
    out: a dict of result arrays of an earlier call with the same shapes ({"parent", "depth", "order"}) to write into again --
    the library overwrites every entry of parent / depth, so nothing is pre-filled (filling two fresh 40 MB arrays with
    CZ_NONE cost 10 ms of a 16 ms call on the 10M-node graph), and reused pages are not faulted in again."""
    on = out_off if isinstance(out_off, DeviceGraph) else None
    if on is None:
        out_off, out_tgt = _csr32(out_off, out_tgt)
    N = on.n if on is not None else out_off.size - 1
    starts = _u32(starts)
    g = _u32(goals) if goals is not None else None

    def result(name, wanted):
        if not wanted:
            return None
        a = (out or {}).get(name)
        if a is None or a.shape != (starts.size, N) or a.dtype != np.uint32 or not a.flags.c_contiguous:
            a = np.empty((starts.size, N), dtype=np.uint32)
        if out is not None:
            out[name] = a
        return a
    parent, depth, order = result("parent", True), result("depth", want_depth), result("order", want_order)
    reached = np.zeros(starts.size, dtype=np.uint32)
    tail = (ptr(starts), starts.size, ptr(g), 0 if g is None else g.size, int(share_visited), ptr(parent), ptr(depth), ptr(order),
            ptr(reached), ptr(poison))
    if on is not None:
        check(_lib.lib().cz_bfs_on(on._h, *tail))
    else:
        check(_lib.lib().cz_bfs(ptr(out_off), ptr(out_tgt), N, out_tgt.size, *tail))
    if order is not None:  # the library writes the discovery order of the nodes it reached; the rest reads CZ_NONE
        for si in range(starts.size):
            order[si, int(reached[si]):] = CZ_NONE
    return parent, depth, order, reached


def connected_components(off, tgt=None, poison=None):
    """cz_connected_components on the symmetrised CSR; `off` may be a DeviceGraph: cz_connected_components_on"""
    k = C.c_uint32(0)
    if isinstance(off, DeviceGraph):
        grp = np.empty(off.n, dtype=np.uint32)
        check(_lib.lib().cz_connected_components_on(off._h, ptr(grp), C.byref(k), ptr(poison)))
        return grp, k.value
    off, tgt = _csr32(off, tgt)
    N = off.size - 1
    grp = np.empty(N, dtype=np.uint32)
    check(_lib.lib().cz_connected_components(ptr(off), ptr(tgt), N, tgt.size, ptr(grp), C.byref(k), ptr(poison)))
    return grp, k.value


def clustering_coefficients(off, tgt, poison=None, symmetric=False):
    """cz_clustering_coefficients on the symmetrised out-CSR -> (n_triangles u64 [N], degree u32 [N]).
    symmetric=True: the caller built the adjacency with as_directed_graph(undirected=True) and vouches for it (CZ_ADJ_SYMMETRIC);
    otherwise the library verifies the symmetry exactly before it takes the kernel that relies on it."""
    off, tgt = _csr32(off, tgt)
    N = off.size - 1
    tri = np.zeros(N, dtype=np.uint64)
    deg = np.zeros(N, dtype=np.uint32)
    check(_lib.lib().cz_clustering_coefficients(ptr(off), ptr(tgt), N, tgt.size, ptr(tri), ptr(deg), ptr(poison),
                                                _lib.CZ_ADJ_SYMMETRIC if symmetric else 0))
    return tri, deg


def sssp(out_off, out_tgt, weights, starts, poison=None, out=None, goals=None):
    """cz_sssp; `out_off` may be a DeviceGraph uploaded with weights (then out_tgt / weights are ignored): cz_sssp_on.
    out: a dict holding the ("dist", "parent") arrays of an earlier call with the same shapes, to be written into again (like bfs)
    goals: stop once every goal of every start is settled (cz_sssp_goals: dijkstra's early exit); nodes not settled by then read
    unreached"""
    g = _u32(goals) if goals is not None else None
    if isinstance(out_off, DeviceGraph):
        starts = _u32(starts)
        shape = (starts.size, out_off.n)
        dist = out.get("dist") if out is not None and getattr(out.get("dist"), "shape", None) == shape else np.empty(shape, dtype=np.float32)
        parent = out.get("parent") if out is not None and getattr(out.get("parent"), "shape", None) == shape else np.empty(shape, dtype=np.uint32)
        if out is not None:
            out["dist"], out["parent"] = dist, parent
        if g is not None:
            check(_lib.lib().cz_sssp_goals_on(out_off._h, ptr(starts), starts.size, ptr(g), g.size, ptr(dist), ptr(parent), ptr(poison)))
        else:
            check(_lib.lib().cz_sssp_on(out_off._h, ptr(starts), starts.size, ptr(dist), ptr(parent), ptr(poison)))
        return dist, parent
    out_off, out_tgt = _csr32(out_off, out_tgt)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    N = out_off.size - 1
    starts = _u32(starts)
    dist = np.empty((starts.size, N), dtype=np.float32)
    parent = np.empty((starts.size, N), dtype=np.uint32)
    if g is not None:
        check(_lib.lib().cz_sssp_goals(ptr(out_off), ptr(out_tgt), ptr(w), N, out_tgt.size, ptr(starts), starts.size, ptr(g), g.size,
                                       ptr(dist), ptr(parent), ptr(poison)))
    else:
        check(_lib.lib().cz_sssp(ptr(out_off), ptr(out_tgt), ptr(w), N, out_tgt.size, ptr(starts), starts.size, ptr(dist),
                                 ptr(parent), ptr(poison)))
    return dist, parent


def random_access_probe(n_words: int, word_bytes: int, n_access: int = 0, reps: int = 0):
    """cz_random_access_probe: (random loads, random atomicMin) in 1e9 accesses / s over a per-node array of this shape"""
    a, b = C.c_double(0.0), C.c_double(0.0)
    check(_lib.lib().cz_random_access_probe(int(n_words), int(word_bytes), int(n_access), int(reps), C.byref(a), C.byref(b)))
    return a.value, b.value


def last_timing():
    """cz_graph_last_timing: (upload_ms, device_ms, download_ms) of this thread's last whole-graph rule call"""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    check(_lib.lib().cz_graph_last_timing(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def closeness(out_off, out_tgt, weights, poison=None):
    """cz_closeness on the weighted out-CSR (weights >= 0) -> centrality f64 [N] (the reference's f32 arithmetic)"""
    out_off, out_tgt = _csr32(out_off, out_tgt)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    N = out_off.size - 1
    cent = np.zeros(N, dtype=np.float64)
    check(_lib.lib().cz_closeness(ptr(out_off), ptr(out_tgt), ptr(w), N, out_tgt.size, ptr(cent), ptr(poison)))
    return cent


def betweenness(out_off, out_tgt, weights, poison=None):
    """cz_betweenness on the weighted out-CSR (weights > 0) -> centrality f64 [N]"""
    out_off, out_tgt = _csr32(out_off, out_tgt)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    N = out_off.size - 1
    cent = np.zeros(N, dtype=np.float64)
    check(_lib.lib().cz_betweenness(ptr(out_off), ptr(out_tgt), ptr(w), N, out_tgt.size, ptr(cent), ptr(poison)))
    return cent


def label_propagation(out_off, out_tgt, weights, max_iter=10, poison=None, symmetric=False):
    """cz_label_propagation on the weighted out-CSR -> (labels u32 [N], iterations run, colour classes).
    symmetric=True: the caller vouches that the adjacency is symmetric (undirected = true); otherwise the library finds out"""
    out_off, out_tgt = _csr32(out_off, out_tgt)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    N = out_off.size - 1
    labels = np.empty(N, dtype=np.uint32)
    it, nc = C.c_uint32(0), C.c_uint32(0)
    check(_lib.lib().cz_label_propagation(ptr(out_off), ptr(out_tgt), ptr(w), N, out_tgt.size, int(max_iter), ptr(labels),
                                          C.byref(it), C.byref(nc), ptr(poison), _lib.CZ_ADJ_SYMMETRIC if symmetric else 0))
    return labels, it.value, nc.value


def debug_seq_sum(rows, init, lanes=64, per_lane=16):
    """TEST HOOK (cz_debug_seq_sum): init[r] + rows[r][0] + rows[r][1] + ... one after the other in f32, by csrc/exact_sum.h's
    wave procedure; rows = a list of float32 arrays."""
    rows = [np.ascontiguousarray(r, dtype=np.float32) for r in rows]
    off = np.zeros(len(rows) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([r.size for r in rows])
    terms = np.concatenate(rows) if rows and off[-1] else np.zeros(1, dtype=np.float32)
    init = np.ascontiguousarray(init, dtype=np.float32)
    out = np.empty(len(rows), dtype=np.float32)
    check(_lib.lib().cz_debug_seq_sum(ptr(terms), ptr(off), ptr(init), len(rows), int(lanes), int(per_lane), ptr(out)))
    return out

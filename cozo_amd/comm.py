"""RCCL communicators behind the C ABI (include/cozo_gpu.h, multi-GPU section): what a Rust `impl FixedRule` binds for the
multi-GPU form of the rules.  One process per GPU here (the launch contract of bench.py / torch.distributed); the 128-byte
unique id travels over whatever the launcher offers -- `Comm.from_torch_distributed` uses the existing process group.
The single-process form (one cozo process driving n GPUs) is `pagerank_multi`."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, ptr


class Comm:
    def __init__(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == _lib.CZ_UNIQUE_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_uint8 * _lib.CZ_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        check(_lib.lib().cz_comm_create_rank(buf, rank, world, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * _lib.CZ_UNIQUE_ID_BYTES)()
        check(_lib.lib().cz_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device=None, group=None):
        """rank 0 draws the id, the process group (any backend) carries it"""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0, group=group)
        return cls(ident[0], rank, world)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cz_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather(self, buf, bytes_per_rank: int, stream: int = 0):
        check(_lib.lib().cz_comm_all_gather(self._h, ptr(buf), bytes_per_rank, C.c_void_p(stream)))

    def all_reduce_sum_f64(self, buf, n: int, stream: int = 0):
        check(_lib.lib().cz_comm_all_reduce_sum_f64(self._h, ptr(buf), n, C.c_void_p(stream)))

    def pagerank_sharded(self, plan, rows_per_rank: int, tolerance: float, max_iter: int, allreduce_exchange: bool = False,
                         poison: Optional[np.ndarray] = None, stream: int = 0):
        """graph::page_rank over row shards, collectively (cz_pagerank_sharded): `plan` is this rank's PageRankPlan over rows
        [rank * rows_per_rank, ...).  Returns (iterations, final error); the rank's scores stay in the plan."""
        it, err = C.c_uint32(0), C.c_double(0.0)
        check(_lib.lib().cz_pagerank_sharded(self._h, plan._h, rows_per_rank, float(tolerance), int(max_iter),
                                             _lib.CZ_PR_EXCHANGE_ALLREDUCE if allreduce_exchange else 0, C.byref(it),
                                             C.byref(err), ptr(poison), C.c_void_p(stream)))
        return it.value, err.value

    def pagerank_sharded_overlapped(self, plan_first, plan_second, rows_per_rank: int, half_rows: int, tolerance: float, max_iter: int,
                                    poison: Optional[np.ndarray] = None, stream: int = 0):
        """cz_pagerank_sharded_overlapped: the rank's rows as two PageRankPlans (rows [rb, rb + half_rows) and the rest); the
        first part's contributions travel while the second part is swept.  Scores equal pagerank_sharded's."""
        it, err = C.c_uint32(0), C.c_double(0.0)
        check(_lib.lib().cz_pagerank_sharded_overlapped(self._h, plan_first._h, plan_second._h, rows_per_rank, half_rows,
                                                        float(tolerance), int(max_iter), C.byref(it), C.byref(err), ptr(poison),
                                                        C.c_void_p(stream)))
        return it.value, err.value

    def hnsw_search_sharded(self, shard_index, queries_dev, B: int, k: int, ef: int, id_offset: int, out_ids, out_dist,
                            out_count, stream: int = 0):
        """hnsw_knn over one sub-index per rank (cz_hnsw_search_sharded): rank 0's queries are broadcast, per-shard lists
        all-gathered and merged.  out_ids int64/uint64 [B][k] (global ids, -1 = empty), out_dist f64, out_count i32 -- device."""
        check(_lib.lib().cz_hnsw_search_sharded(self._h, shard_index._h, ptr(queries_dev), B, k, ef, int(id_offset),
                                                ptr(out_ids), ptr(out_dist), ptr(out_count), C.c_void_p(stream)))


def _shard_csr(off_local, tgt):
    off_local = np.ascontiguousarray(off_local, dtype=np.uint32)
    tgt = np.ascontiguousarray(tgt, dtype=np.uint32)
    return off_local, tgt


def bfs_sharded(comm: Comm, off_local, tgt, n: int, row_begin: int, row_end: int, starts, goals=None, share_visited=False,
                want_depth=False, want_order=False, poison=None):
    """ONE BFS over a vertex-partitioned graph, collectively (cz_bfs_sharded): this rank passes the out-adjacency of
    [row_begin, row_end); outputs as cozo_amd.graph.bfs, identical on every rank and to the whole graph's on one GPU."""
    off_local, tgt = _shard_csr(off_local, tgt)
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    g = None if goals is None else np.ascontiguousarray(goals, dtype=np.uint32)
    parent = np.full((starts.size, n), _lib.CZ_NONE, dtype=np.uint32)
    depth = np.full((starts.size, n), _lib.CZ_NONE, dtype=np.uint32) if want_depth else None
    order = np.full((starts.size, n), _lib.CZ_NONE, dtype=np.uint32) if want_order else None
    reached = np.zeros(starts.size, dtype=np.uint32)
    check(_lib.lib().cz_bfs_sharded(comm._h, ptr(off_local), ptr(tgt), n, row_begin, row_end, tgt.size, ptr(starts), starts.size,
                                    ptr(g), 0 if g is None else g.size, int(share_visited), ptr(parent), ptr(depth), ptr(order),
                                    ptr(reached), ptr(poison)))
    return parent, depth, order, reached


def connected_components_sharded(comm: Comm, off_local, tgt, n: int, row_begin: int, row_end: int, poison=None):
    """ConnectedComponents over a vertex partition of the symmetrised graph, collectively (cz_connected_components_sharded)
    -> (group u32 [n], n_groups, rounds): the same rows on every rank, the single-GPU rule's numbering"""
    off_local, tgt = _shard_csr(off_local, tgt)
    grp = np.empty(n, dtype=np.uint32)
    k, rounds = C.c_uint32(0), C.c_uint32(0)
    check(_lib.lib().cz_connected_components_sharded(comm._h, ptr(off_local), ptr(tgt), n, row_begin, row_end, tgt.size, ptr(grp),
                                                     C.byref(k), C.byref(rounds), ptr(poison)))
    return grp, k.value, rounds.value


def sssp_sharded(comm: Comm, off_local, tgt, weights, n: int, row_begin: int, row_end: int, starts, poison=None):
    """ONE SSSP per start over a vertex-partitioned graph, collectively (cz_sssp_sharded) -> (dist f32, parent) [starts][n]"""
    off_local, tgt = _shard_csr(off_local, tgt)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    dist = np.empty((starts.size, n), dtype=np.float32)
    parent = np.empty((starts.size, n), dtype=np.uint32)
    check(_lib.lib().cz_sssp_sharded(comm._h, ptr(off_local), ptr(tgt), ptr(w), n, row_begin, row_end, tgt.size, ptr(starts),
                                     starts.size, ptr(dist), ptr(parent), ptr(poison)))
    return dist, parent


def sssp_sharded_last_stats():
    """cz_sssp_sharded_last_stats: dict(rounds, pairs, compactions, buckets) of this thread's last cz_sssp_sharded call"""
    out = np.zeros(4, dtype=np.uint64)
    check(_lib.lib().cz_sssp_sharded_last_stats(ptr(out)))
    return dict(rounds=int(out[0]), pairs=int(out[1]), compactions=int(out[2]), buckets=int(out[3]))


def pagerank_multi(in_off, in_src, out_deg, n_gpus: int, damping=0.85, tolerance=1e-4, max_iter=10,
                   allreduce_exchange=False, overlap_exchange=False, poison=None):
    """cz_pagerank on n_gpus devices of THIS process (one host thread + one RCCL communicator per GPU)."""
    in_off = np.ascontiguousarray(in_off, dtype=np.uint32)
    in_src = np.ascontiguousarray(in_src, dtype=np.uint32)
    out_deg = np.ascontiguousarray(out_deg, dtype=np.uint32)
    N = out_deg.size
    scores = np.empty(N, dtype=np.float32)
    it, err = C.c_uint32(0), C.c_double(0.0)
    flags = (_lib.CZ_PR_EXCHANGE_ALLREDUCE if allreduce_exchange else 0) | (_lib.CZ_PR_OVERLAP_EXCHANGE if overlap_exchange else 0)
    check(_lib.lib().cz_pagerank_multi(ptr(in_off), ptr(in_src), ptr(out_deg), N, in_src.size, np.float32(damping),
                                       float(tolerance), int(max_iter), int(n_gpus), flags, ptr(scores), C.byref(it),
                                       C.byref(err), ptr(poison)))
    return scores, it.value, err.value


def bfs_multi(out_off, out_tgt, n_gpus: int, starts, goals=None, share_visited=False, want_depth=False, want_order=False, poison=None):
    """cz_bfs on n_gpus devices of THIS process (cz_bfs_multi): the whole host CSR in, cz_bfs's results out"""
    out_off = np.ascontiguousarray(out_off, dtype=np.uint32)
    out_tgt = np.ascontiguousarray(out_tgt, dtype=np.uint32)
    n = out_off.size - 1
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    g = np.ascontiguousarray(goals, dtype=np.uint32) if goals is not None else None
    parent = np.empty((starts.size, n), dtype=np.uint32)
    depth = np.empty((starts.size, n), dtype=np.uint32) if want_depth else None
    order = np.full((starts.size, n), _lib.CZ_NONE, dtype=np.uint32) if want_order else None
    reached = np.zeros(starts.size, dtype=np.uint32)
    check(_lib.lib().cz_bfs_multi(ptr(out_off), ptr(out_tgt), n, out_tgt.size, int(n_gpus), ptr(starts), starts.size, ptr(g),
                                  0 if g is None else g.size, int(share_visited), ptr(parent), ptr(depth), ptr(order), ptr(reached),
                                  ptr(poison)))
    return parent, depth, order, reached


def sssp_multi(out_off, out_tgt, weights, n_gpus: int, starts, poison=None):
    """cz_sssp on n_gpus devices of this process (cz_sssp_multi)"""
    out_off = np.ascontiguousarray(out_off, dtype=np.uint32)
    out_tgt = np.ascontiguousarray(out_tgt, dtype=np.uint32)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    n = out_off.size - 1
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    dist = np.empty((starts.size, n), dtype=np.float32)
    parent = np.empty((starts.size, n), dtype=np.uint32)
    check(_lib.lib().cz_sssp_multi(ptr(out_off), ptr(out_tgt), ptr(w), n, out_tgt.size, int(n_gpus), ptr(starts), starts.size, ptr(dist),
                                   ptr(parent), ptr(poison)))
    return dist, parent


def connected_components_multi(off, tgt, n_gpus: int, poison=None):
    """cz_connected_components on n_gpus devices of this process (cz_connected_components_multi)"""
    off = np.ascontiguousarray(off, dtype=np.uint32)
    tgt = np.ascontiguousarray(tgt, dtype=np.uint32)
    n = off.size - 1
    grp = np.empty(n, dtype=np.uint32)
    k = C.c_uint32(0)
    check(_lib.lib().cz_connected_components_multi(ptr(off), ptr(tgt), n, tgt.size, int(n_gpus), ptr(grp), C.byref(k), ptr(poison)))
    return grp, k.value


class HnswMulti:
    """An index partitioned over n GPUs of THIS process (cz_hnsw_multi_*): sub-index r on GPU r, built there from the rows
    [r * ceil(n / n_gpus), ...) of `vectors`; a search runs on every device at once and merges the per-shard lists by
    (distance, id).  Ids are GLOBAL row numbers."""

    def __init__(self, handle, n_gpus: int):
        self._h, self.n_gpus = handle, n_gpus

    @classmethod
    def build(cls, manifest, vectors, n_gpus: int, seed: int = 0, max_batch: int = 0):
        from .hnsw import DISTANCES, _build_flags
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        h = C.c_void_p()
        nd = C.c_uint64(0)
        check(_lib.lib().cz_hnsw_multi_build(ptr(v), v.shape[0], manifest.vec_dim, DISTANCES[manifest.distance], manifest.m_neighbours,
                                             manifest.ef_construction, int(manifest.keep_pruned_connections), int(seed), int(max_batch),
                                             int(n_gpus), _build_flags(manifest), C.byref(nd), C.byref(h)))
        self = cls(h, n_gpus)
        self.build_n_dist = nd.value
        return self

    def id_offsets(self) -> np.ndarray:
        off = np.zeros(self.n_gpus, dtype=np.uint64)
        _lib.lib().cz_hnsw_multi_shards(self._h, ptr(off))
        return off

    def search(self, queries, k: int, ef: int):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        B = q.shape[0]
        ids = np.empty((B, k), dtype=np.uint64)
        dist = np.empty((B, k), dtype=np.float64)
        cnt = np.empty(B, dtype=np.uint32)
        check(_lib.lib().cz_hnsw_multi_search(self._h, ptr(q), B, k, ef, ptr(ids), ptr(dist), ptr(cnt)))
        return ids, dist, cnt

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cz_hnsw_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Multi-GPU form of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI;
"gloo" on CPU for the tests).  Only the two real exchange steps are collectives:

* PageRank: 1-D row (destination-node) partition of the in-CSR.  Every rank keeps the full contribution
  vector, computes its own rows with the plan kernels, then ALL-GATHERS its slice of the next contribution
  vector (N/world * 4 bytes per rank per iteration, one direct hop per peer on the xGMI mesh) and ALL-REDUCES
  one f64 (the |delta| sum that drives the reference's stopping rule).  A ring all-reduce of a zero-padded
  full vector would move ~2x(world-1)/world * 4N bytes per link instead -- not used.
* HNSW over an index partitioned into independent sub-indices (one per rank): every rank searches its own
  shard for the same query batch, the per-shard (distance, id) lists are all-gathered (B*k*12 bytes per rank)
  and merged to the global top-k on every rank.

Query batches that are independent units are simply split across ranks (no collective): that is what
bench.py does for the HNSW scaling line.

The local compute is passed in as a callable so that the exchange logic can be exercised with world_size 2
on CPU (gloo) in tests/; the product path binds it to the GPU kernels (cozo_amd.graph.PageRankPlan /
cozo_amd.hnsw.GpuHnswIndex).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def equal_row_partition(n_rows: int, world: int) -> Tuple[int, list]:
    """rows per rank (padded so every rank owns the same number: all_gather needs equal slices) and the
    [begin, end) ranges clipped to n_rows."""
    per = (n_rows + world - 1) // world
    return per, [(min(n_rows, r * per), min(n_rows, (r + 1) * per)) for r in range(world)]


class ShardedPageRank:
    """graph::page_rank (pagerank.rs:47-50) over row shards.

    local_init(contrib_full)                      -> writes init/out_degree for ALL nodes, resets local scores
    local_step(contrib_in, contrib_out, err)      -> computes this rank's rows from the full contrib_in, writes
                                                     ITS slice [rank*per, (rank+1)*per) of the full-length
                                                     contrib_out and adds its sum |new - old| into err (f64[1])
    """

    def __init__(self, n_nodes: int, rank: int, world: int, device: torch.device,
                 local_init: Callable, local_step: Callable, group=None):
        self.n, self.rank, self.world, self.device, self.group = n_nodes, rank, world, device, group
        self.per, self.ranges = equal_row_partition(n_nodes, world)
        self.local_init, self.local_step = local_init, local_step
        padded = self.per * world
        self.contrib = [torch.zeros(padded, dtype=torch.float32, device=device) for _ in range(2)]
        self.err = torch.zeros(1, dtype=torch.float64, device=device)

    def run(self, tolerance: float, max_iter: int, poison: Optional[Callable[[], bool]] = None):
        """Returns (iterations, final error).  Stopping rule of graph::page_rank: err < tolerance or
        iterations == max_iter, decided on the all-reduced error so that every rank stops together."""
        cin, cout = self.contrib
        self.local_init(cin)
        rb = self.rank * self.per
        it = 0
        # `err < tolerance` can never hold for tolerance <= 0 (err is a sum of absolute values): the loop then runs
        # max_iter sweeps back to back without a host round trip per iteration; the error is read once at the end
        never_stops_early = not (tolerance > 0.0)
        # Cancellation is collective (like the C++ loop behind the C ABI, csrc/sharded_pagerank.hpp): a rank whose poison
        # flag is set stops sweeping but keeps taking part in the exchanges, its flag travels with the error in the same
        # all-reduce, and every rank raises at the same iteration -- a rank that raised on its own would leave the others
        # blocked in the next collective.
        err2 = torch.zeros(2, dtype=torch.float64, device=self.device)
        while True:
            p = poison is not None and bool(poison())
            last = it + 1 == max_iter
            self.err.zero_()
            if not p:
                self.local_step(cin, cout, self.err)
            look = last or not never_stops_early or (poison is not None and (it + 1) % 8 == 0)
            if self.world > 1:
                # in-place all-gather: this rank's slice already sits at its final position in `cout`
                dist.all_gather_into_tensor(cout, cout[rb:rb + self.per], group=self.group)
                if look:
                    err2[0] = self.err[0]
                    err2[1] = 1.0 if p else 0.0
                    dist.all_reduce(err2, op=dist.ReduceOp.SUM, group=self.group)
                    self.err[0] = err2[0]
                    p = bool(err2[1].item() > 0)
            cin, cout = cout, cin
            it += 1
            if p and (look or self.world == 1):
                raise RuntimeError("ProcessKilled")
            if never_stops_early and not last:
                continue
            e = float(self.err.item())
            if e < tolerance or it == max_iter:
                self.contrib = [cin, cout]  # contrib[0] is the vector the next sweep would read
                return it, e


def merge_shard_topk(local_ids: torch.Tensor, local_dist: torch.Tensor, id_offset: int, k: int, world: int,
                     group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Global top-k of per-shard k-NN lists.  local_ids [B][k] (0xFFFFFFFF = empty, as int64 here),
    local_dist [B][k] f64; ids are made global by adding id_offset.  Every rank gets the same merged
    (ids [B][k] int64 with -1 for empty, dist [B][k]) ordered by (distance, id) -- the order hnsw_knn returns."""
    ids = local_ids.to(torch.int64)
    empty = ids == 0xFFFFFFFF
    ids = torch.where(empty, torch.full_like(ids, -1), ids + id_offset)
    d = torch.where(empty, torch.full_like(local_dist, float("inf")), local_dist)
    if world > 1:
        B = ids.shape[0]
        all_ids = torch.empty((world * B, k), dtype=ids.dtype, device=ids.device)
        all_d = torch.empty((world * B, k), dtype=d.dtype, device=d.device)
        dist.all_gather_into_tensor(all_ids, ids.contiguous(), group=group)
        dist.all_gather_into_tensor(all_d, d.contiguous(), group=group)
        ids = all_ids.view(world, B, k).permute(1, 0, 2).reshape(B, world * k)
        d = all_d.view(world, B, k).permute(1, 0, 2).reshape(B, world * k)
    # order by (distance, id): stable sort on id first, then on distance (NaN sorts last like ordered-float)
    big = torch.iinfo(torch.int64).max
    key_ids = torch.where(ids < 0, torch.full_like(ids, big), ids)
    o1 = torch.argsort(key_ids, dim=1, stable=True)
    ids, d = torch.gather(ids, 1, o1), torch.gather(d, 1, o1)
    o2 = torch.argsort(d, dim=1, stable=True)
    ids, d = torch.gather(ids, 1, o2)[:, :k], torch.gather(d, 1, o2)[:, :k]
    return ids, d


def symmetrised_csr(n: int, a, b):
    """out-CSR (offsets u64 [n+1], targets u32) of the rows (a,b) and their mirrors, neighbour lists ascending,
    duplicates kept -- the layout as_directed_graph(undirected = true) hands to the rules (fixed_rule/mod.rs:187-195)."""
    import numpy as np
    a = np.asarray(a, dtype=np.uint32)
    b = np.asarray(b, dtype=np.uint32)
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a])
    order = np.lexsort((dst, src)) if src.size else np.zeros(0, dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.uint64)
    if src.size:
        off[1:] = np.cumsum(np.bincount(src, minlength=n))
    return off, dst[order].astype(np.uint32)


def sharded_connected_components(n_nodes: int, edge_from, edge_to, world: int, device: torch.device,
                                 local_cc: Callable, group=None):
    """ConnectedComponents (strongly_connected_components.rs:42-77, strong = false) over an EDGE partition: every rank
    holds any subset of the edge rows (dense ids over the same n_nodes), and the union over ranks is the relation.

    1. local_cc(offsets, targets) -> (group u32 [N], n_groups) on the rank's own symmetrised rows (the single-GPU rule:
       group = rank of the component by its smallest member).  label[v] = smallest member of v's local component.
    2. ONE exchange: all-gather of the label vectors (4N bytes per rank).
    3. the components of the union are the components of the star graph {(v, label_r[v]) : r < world, label_r[v] != v}
       (<= world * N rows): every rank runs local_cc on it.  Replicated on purpose: the result is needed everywhere and
       a broadcast of 4N bytes costs what the gather did.
    Group ids come out exactly as the single-process rule numbers them (ranked by smallest member), so the rows are
    identical to the unsharded run's.  Returns (group u32 [N], n_groups)."""
    import numpy as np
    off, tgt = symmetrised_csr(n_nodes, edge_from, edge_to)
    grp, _ = local_cc(off, tgt)
    grp = np.asarray(grp, dtype=np.int64)
    # groups are numbered in order of their smallest member, so the first node carrying a group id IS that member
    _, first = np.unique(grp, return_index=True)
    label = first[grp].astype(np.int64) if n_nodes else np.zeros(0, dtype=np.int64)
    if world > 1:
        mine = torch.from_numpy(label).to(device)
        every = torch.empty(world * n_nodes, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(every, mine, group=group)
        labels = every.cpu().numpy().reshape(world, n_nodes)
    else:
        labels = label.reshape(1, n_nodes)
    v = np.broadcast_to(np.arange(n_nodes, dtype=np.int64), labels.shape)
    keep = labels != v
    off2, tgt2 = symmetrised_csr(n_nodes, v[keep], labels[keep])
    grp2, k = local_cc(off2, tgt2)
    return np.asarray(grp2, dtype=np.uint32), int(k)


def shard_sources(n_sources: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the start nodes this rank traverses from.  Traversals from different start nodes are
    independent units (ShortestPathBFS / ShortestPathDijkstra run them one after another or under rayon,
    shortest_path_dijkstra.rs:70-153; ClosenessCentrality is one SSSP per node, all_pairs_shortest_path.rs:113-144):
    they are split across ranks with NO collective on the data path."""
    per = (n_sources + world - 1) // world
    return min(n_sources, rank * per), min(n_sources, (rank + 1) * per)


def gather_source_rows(local_rows: torch.Tensor, n_sources: int, world: int, group=None) -> torch.Tensor:
    """Puts the per-source result rows ([local sources][width], any dtype) of every rank back into start-node order on
    every rank: one all-gather of the padded slices at the END of the job (not part of the traversal)."""
    if world == 1:
        return local_rows
    per = (n_sources + world - 1) // world
    width = local_rows.shape[1:]
    mine = torch.zeros((per,) + tuple(width), dtype=local_rows.dtype, device=local_rows.device)
    mine[:local_rows.shape[0]] = local_rows
    every = torch.empty((world * per,) + tuple(width), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(every, mine.contiguous(), group=group)
    return every[:n_sources]


def sharded_hnsw_knn(local_search: Callable, queries: Optional[torch.Tensor], n_queries: int, dim: int, id_offset: int, k: int,
                     rank: int, world: int, device: torch.device, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """hnsw_knn over an index partitioned into one independent sub-index per rank (config 4 of BASELINE.json: 10 M x 768
    as 8 shards of 1.25 M): rank 0 holds the query batch and broadcasts it (B x dim x 4 bytes), every rank searches ITS
    shard with the same k / ef through `local_search(queries f32 [B][dim]) -> (ids [B][k] with 0xFFFFFFFF padding, dist f64
    [B][k])` (bound to GpuHnswIndex.hnsw_knn_batch on the device), and merge_shard_topk all-gathers and merges the lists.
    Every rank returns the same (ids int64 [B][k], -1 padded, global = local + id_offset; dist [B][k]).  Recall of the
    merged result is at least the per-shard recall: a true neighbour is in exactly one shard and competes there with fewer
    candidates."""
    if world > 1:
        q = queries.to(device=device, dtype=torch.float32).contiguous() if rank == 0 else \
            torch.empty((n_queries, dim), dtype=torch.float32, device=device)
        dist.broadcast(q, src=0, group=group)
    else:
        q = queries.to(device=device, dtype=torch.float32)
    ids, d = local_search(q)
    return merge_shard_topk(torch.as_tensor(ids).to(device), torch.as_tensor(d).to(device), id_offset, k, world, group=group)


class OverlappedShardedPageRank:
    """ShardedPageRank with the exchange of the first half of a rank's rows in flight while the second half computes.

    The exchange of iteration k feeds iteration k+1, so the only overlap there is lies inside an iteration: the rank's row
    range [rb, re) is cut at `mid` into TWO plans (two cz_pagerank_plan handles over the two sub-CSRs -- no new device code);
    step(first) -> start gathering every rank's first-half slice (async, on the collective's own stream) -> step(second) runs
    meanwhile -> gather the second-half slices -> wait.  Slices land at their natural places in the full contribution vector
    (a list all-gather into views), so sources keep their ids and every row sum keeps its order: scores are bit-identical to
    the unsplit run.  Hides min(second-half sweep, first-half gather) per iteration.

    local_init(contrib_full); local_steps = (step_first, step_second), each (contrib_in, contrib_out, err) like
    ShardedPageRank.local_step but for rows [rb, mid) / [mid, re).  `halves` = how many rows of the padded per-rank range
    belong to the first plan (the same on every rank)."""

    def __init__(self, n_nodes: int, rank: int, world: int, device: torch.device, local_init: Callable,
                 local_steps: Tuple[Callable, Callable], halves: Optional[int] = None, group=None):
        self.n, self.rank, self.world, self.device, self.group = n_nodes, rank, world, device, group
        self.per, self.ranges = equal_row_partition(n_nodes, world)
        self.half = self.per // 2 if halves is None else halves
        self.local_init, self.local_steps = local_init, local_steps
        padded = self.per * world
        self.contrib = [torch.zeros(padded, dtype=torch.float32, device=device) for _ in range(2)]
        self.err = torch.zeros(1, dtype=torch.float64, device=device)

    def split_rows(self) -> Tuple[int, int, int]:
        """(row_begin, mid, row_end) of this rank: what the two plans are created over"""
        rb, re = self.ranges[self.rank]
        return rb, min(re, self.rank * self.per + self.half), re

    def _views(self, buf: torch.Tensor, first: bool):
        lo, hi = (0, self.half) if first else (self.half, self.per)
        return [buf[r * self.per + lo:r * self.per + hi] for r in range(self.world)]

    def run(self, tolerance: float, max_iter: int, poison: Optional[Callable[[], bool]] = None):
        cin, cout = self.contrib
        self.local_init(cin)
        it = 0
        never_stops_early = not (tolerance > 0.0)
        err2 = torch.zeros(2, dtype=torch.float64, device=self.device)
        while True:
            p = poison is not None and bool(poison())  # collective cancellation: see ShardedPageRank.run
            last = it + 1 == max_iter
            look = last or not never_stops_early or (poison is not None and (it + 1) % 8 == 0)
            self.err.zero_()
            if not p:
                self.local_steps[0](cin, cout, self.err)
            pending = None
            if self.world > 1 and self.half > 0:
                v = self._views(cout, True)
                pending = dist.all_gather(v, v[self.rank], group=self.group, async_op=True)
            if not p:
                self.local_steps[1](cin, cout, self.err)
            if self.world > 1:
                if self.per - self.half > 0:
                    v = self._views(cout, False)
                    dist.all_gather(v, v[self.rank], group=self.group)
                if pending is not None:
                    pending.wait()
                if look:
                    err2[0] = self.err[0]
                    err2[1] = 1.0 if p else 0.0
                    dist.all_reduce(err2, op=dist.ReduceOp.SUM, group=self.group)
                    self.err[0] = err2[0]
                    p = bool(err2[1].item() > 0)
            cin, cout = cout, cin
            it += 1
            if p and (look or self.world == 1):
                raise RuntimeError("ProcessKilled")
            if never_stops_early and not last:
                continue
            e = float(self.err.item())
            if e < tolerance or it == max_iter:
                self.contrib = [cin, cout]
                return it, e

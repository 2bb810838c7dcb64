"""The N > 1 form of the hot path (cozo_amd/distributed.py), world_size 2 over gloo on CPU.

The exchange logic (row partition, in-place all-gather of the contribution slice, all-reduced stopping rule;
all-gather + merge of per-shard k-NN lists) is what is under test; the local compute is a numpy restatement of
one row-shard sweep here (on the GPU box the same callables are bound to cz_pagerank_plan_step /
cz_hnsw_search_batch, bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _shard_step_factory(n, ioff, isrc, outdeg, rb, re, damping):
    """one Jacobi sweep over rows [rb, re): sequential f32 sums in sorted in-neighbour order (graph::page_rank)"""
    d = np.float32(damping)
    base = (np.float32(1.0) - d) / np.float32(n)
    init = np.float32(1.0) / np.float32(n)
    state = {"scores": np.full(re - rb, init, dtype=np.float32)}

    def local_init(contrib):
        with np.errstate(divide="ignore"):
            contrib[:n] = torch.from_numpy((np.full(n, init, dtype=np.float32) / outdeg.astype(np.float32)))
        state["scores"][:] = init

    def local_step(cin, cout, err):
        c = cin.numpy()
        o = cout.numpy()
        e = 0.0
        for r in range(rb, re):
            s = np.float32(0)
            for k in range(int(ioff[r]), int(ioff[r + 1])):
                s = np.float32(s + c[isrc[k]])
            nw = np.float32(base + np.float32(d * s))
            e += abs(float(np.float32(nw - state["scores"][r - rb])))
            state["scores"][r - rb] = nw
            with np.errstate(divide="ignore"):
                o[r] = nw / np.float32(outdeg[r])
        err += e

    return local_init, local_step, state


def _pagerank_worker(rank, world, port, n, ioff, isrc, outdeg, tol, max_iter, q):
    from cozo_amd.distributed import ShardedPageRank, equal_row_partition
    _init(rank, world, port)
    per, ranges = equal_row_partition(n, world)
    rb, re = ranges[rank]
    li, ls, state = _shard_step_factory(n, ioff, isrc, outdeg, rb, re, 0.85)
    sp = ShardedPageRank(n, rank, world, torch.device("cpu"), li, ls)
    it, err = sp.run(tol, max_iter)
    q.put((rank, rb, re, state["scores"].copy(), it, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,e,tol,max_iter", [(301, 2500, 1e-4, 10), (64, 400, 0.0, 5), (7, 12, 1e-4, 10)])
def test_sharded_pagerank_world2_matches_single_process(oracle, n, e, tol, max_iter):
    frm, to = util.random_relation(n, e, 17)
    g = util.graph_from_relation(oracle, frm, to)
    want, want_it, want_err = oracle.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, max_iter)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pagerank_worker, args=(r, 2, port, g["n"], g["ioff"], g["isrc"], g["outdeg"], tol,
                                                         max_iter, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([r[3] for r in res])
    assert [r[4] for r in res] == [want_it, want_it]  # both ranks stop at the same iteration, same as one process
    assert np.array_equal(got, want)  # row sharding does not change a single bit of the scores
    # the error is a sum of per-rank partial sums: equal up to the f64 reassociation of two partials
    assert abs(res[0][5] - want_err) <= 1e-12 * max(1.0, abs(want_err)) and res[0][5] == res[1][5]


def _topk_worker(rank, world, port, base, queries, k, q):
    from cozo_amd.distributed import merge_shard_topk
    _init(rank, world, port)
    per = (base.shape[0] + world - 1) // world
    lo, hi = rank * per, min(base.shape[0], (rank + 1) * per)
    shard = base[lo:hi].astype(np.float64)
    d = ((queries[:, None, :].astype(np.float64) - shard[None, :, :]) ** 2).sum(-1)  # [B][n_shard]
    kk = min(k, hi - lo)
    idx = np.argsort(d, axis=1, kind="stable")[:, :kk]
    ids = np.full((queries.shape[0], k), 0xFFFFFFFF, dtype=np.int64)
    dd = np.full((queries.shape[0], k), np.inf)
    ids[:, :kk] = idx
    dd[:, :kk] = np.take_along_axis(d, idx, 1)
    mi, md = merge_shard_topk(torch.from_numpy(ids), torch.from_numpy(dd), lo, k, world)
    q.put((rank, mi.numpy(), md.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(200, 10), (13, 10)])
def test_shard_topk_merge_world2(n, k):
    rng = np.random.default_rng(5)
    base = rng.standard_normal((n, 16)).astype(np.float32)
    base[3] = base[n - 2]  # an exact distance tie across shards: ordered by id
    queries = rng.standard_normal((9, 16)).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_topk_worker, args=(r, 2, port, base, queries, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = ((queries[:, None, :].astype(np.float64) - base[None, :, :].astype(np.float64)) ** 2).sum(-1)
    order = np.lexsort((np.broadcast_to(np.arange(n), d.shape), d), axis=1)[:, :k]
    for _, mi, md in res:  # every rank holds the same merged result = the global top-k by (distance, id)
        assert np.array_equal(mi[:, :min(k, n)], order[:, :min(k, n)])
        assert np.array_equal(md[:, :min(k, n)], np.take_along_axis(d, order, 1)[:, :min(k, n)])
    assert np.array_equal(res[0][1], res[1][1])


def test_equal_row_partition_covers_rows():
    from cozo_amd.distributed import equal_row_partition
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            per, ranges = equal_row_partition(n, w)
            assert per * w >= n and ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert all(0 <= hi - lo <= per for lo, hi in ranges)


def _cc_worker(rank, world, port, n, fi, ti, q):
    from oracle import oracle as O
    from cozo_amd.distributed import sharded_connected_components
    _init(rank, world, port)
    # an edge partition with nothing nice about it: rows dealt round-robin
    mine = slice(rank, None, world)
    grp, k = sharded_connected_components(n, fi[mine], ti[mine], world, torch.device("cpu"),
                                          lambda off, tgt: O.tarjan_groups(off.size - 1, off, tgt))
    q.put((rank, grp, k))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,e", [(400, 300), (50, 400), (9, 0), (1000, 700)])
def test_sharded_connected_components_world2_matches_single_process(oracle, n, e):
    """edge-partitioned ConnectedComponents: group ids (ranked by smallest member) identical to the unsharded rule's"""
    rng = np.random.default_rng(n + e)
    fi = rng.integers(0, n, e).astype(np.uint32)
    ti = rng.integers(0, n, e).astype(np.uint32)
    off, tgt = oracle.build_csr(n, fi, ti, undirected=True)
    want, want_k = oracle.tarjan_groups(n, off, tgt)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cc_worker, args=(r, 2, port, n, fi, ti, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, grp, k in res:
        assert k == want_k and np.array_equal(grp, want)


def test_sharded_connected_components_world1_and_symmetrised_csr(oracle):
    from cozo_amd.distributed import sharded_connected_components, symmetrised_csr
    rng = np.random.default_rng(2)
    n, e = 120, 90
    fi = rng.integers(0, n, e).astype(np.uint32)
    ti = rng.integers(0, n, e).astype(np.uint32)
    off, tgt = oracle.build_csr(n, fi, ti, undirected=True)
    off2, tgt2 = symmetrised_csr(n, fi, ti)
    assert np.array_equal(off, off2) and np.array_equal(tgt, tgt2)  # same layout as as_directed_graph(undirected)
    want, want_k = oracle.tarjan_groups(n, off, tgt)
    grp, k = sharded_connected_components(n, fi, ti, 1, torch.device("cpu"),
                                          lambda o, t: oracle.tarjan_groups(o.size - 1, o, t))
    assert k == want_k and np.array_equal(grp, want)


def _sources_worker(rank, world, port, n, off, tgt, w, starts, q):
    from oracle import oracle as O
    from cozo_amd.distributed import gather_source_rows, shard_sources
    _init(rank, world, port)
    lo, hi = shard_sources(len(starts), rank, world)
    rows = np.zeros((hi - lo, n), dtype=np.float32)
    for i, s in enumerate(starts[lo:hi]):
        rows[i], _ = O.dijkstra(n, off, tgt, w, int(s))
    every = gather_source_rows(torch.from_numpy(rows), len(starts), world)
    q.put((rank, every.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_starts", [7, 2, 1])
def test_sharded_sources_world2(oracle, n_starts):
    """start nodes are independent units: split across ranks, rows put back in start order by one final gather"""
    frm, to = util.random_relation(60, 300, 3)
    wts = np.random.default_rng(4).random(frm.size).astype(np.float32)
    g = util.graph_from_relation(oracle, frm, to, weights=wts)
    starts = np.arange(n_starts, dtype=np.uint32) * 3
    want = np.stack([oracle.dijkstra(g["n"], g["ooff"], g["otgt"], g["ow"], int(s))[0] for s in starts])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sources_worker, args=(r, 2, port, g["n"], g["ooff"], g["otgt"], g["ow"], starts, q))
             for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got in res:
        assert np.array_equal(got, want)


def _sharded_knn_worker(rank, world, port, base, queries, k, ef, q):
    from oracle import oracle as O
    from cozo_amd.distributed import sharded_hnsw_knn
    _init(rank, world, port)
    per = (base.shape[0] + world - 1) // world
    lo, hi = rank * per, min(base.shape[0], (rank + 1) * per)
    _, flat = util.build_index(O, base[lo:hi], O.L2, 8, 40, seed=rank + 1)  # an independent sub-index per rank

    def local_search(qt):
        ids, dist, _, _ = flat.knn_batch(qt.numpy(), k, ef)
        return ids.astype(np.int64), dist
    mi, md = sharded_hnsw_knn(local_search, torch.from_numpy(queries) if rank == 0 else None, queries.shape[0], queries.shape[1],
                              lo, k, rank, world, torch.device("cpu"))
    q.put((rank, mi.numpy(), md.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hnsw_knn_world2(oracle):
    """two independent sub-indices, queries broadcast from rank 0, lists merged: every rank ends with the same global top-k,
    and its recall against the exact scan is at least what one index over everything reaches"""
    rng = np.random.default_rng(8)
    base = rng.random((1200, 16), dtype=np.float32)
    queries = rng.random((20, 16), dtype=np.float32)
    k, ef = 10, 60
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_knn_worker, args=(r, 2, port, base, queries, k, ef, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    d = ((queries[:, None, :].astype(np.float64) - base[None, :, :].astype(np.float64)) ** 2).sum(-1)
    truth = np.argsort(d, axis=1, kind="stable")[:, :k]
    hits = sum(len(set(res[0][1][i].tolist()) & set(truth[i].tolist())) for i in range(queries.shape[0]))
    assert hits / (k * queries.shape[0]) >= 0.95
    assert np.all(np.diff(res[0][2], axis=1) >= 0)  # ascending by distance


def _overlap_worker(rank, world, port, n, ioff, isrc, outdeg, tol, max_iter, q):
    from cozo_amd.distributed import OverlappedShardedPageRank
    _init(rank, world, port)
    probe = OverlappedShardedPageRank(n, rank, world, torch.device("cpu"), None, (None, None))
    rb, mid, re = probe.split_rows()
    li1, ls1, st1 = _shard_step_factory(n, ioff, isrc, outdeg, rb, mid, 0.85)
    li2, ls2, st2 = _shard_step_factory(n, ioff, isrc, outdeg, mid, re, 0.85)

    def init(c):
        li1(c)
        li2(c)
    sp = OverlappedShardedPageRank(n, rank, world, torch.device("cpu"), init, (ls1, ls2))
    it, err = sp.run(tol, max_iter)
    q.put((rank, rb, re, np.concatenate([st1["scores"], st2["scores"]]), it, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,e,tol,max_iter", [(301, 2500, 1e-4, 10), (64, 400, 0.0, 5), (7, 12, 1e-4, 10), (2, 1, 0.0, 3)])
def test_overlapped_sharded_pagerank_world2_is_bit_identical(oracle, n, e, tol, max_iter):
    """two plans per rank, first-half slices gathered (async) while the second half computes: the scores, the iteration count
    and the stopping decision of the single-process run"""
    frm, to = util.random_relation(n, e, 19)
    g = util.graph_from_relation(oracle, frm, to)
    want, want_it, want_err = oracle.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, max_iter)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, g["n"], g["ioff"], g["isrc"], g["outdeg"], tol, max_iter, q))
             for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([r[3] for r in res])
    assert [r[4] for r in res] == [want_it, want_it]
    assert np.array_equal(got, want)
    assert abs(res[0][5] - want_err) <= 1e-12 * max(1.0, abs(want_err)) and res[0][5] == res[1][5]

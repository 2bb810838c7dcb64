"""The multi-GPU PageRank loop that sits BEHIND the C ABI (cozo_amd/csrc/sharded_pagerank.hpp: what cz_pagerank_sharded and
cz_pagerank_multi run over HIP + RCCL), exercised with world_size 2 on CPU: tests/cpp/sharded_driver_test.cpp instantiates
the same template with a host backend, and the exchange steps run over torch.distributed/gloo through ctypes callbacks.
Scores must be bit-identical to the single-process oracle for both exchange forms; cancellation must be collective."""
import ctypes as C
import os
import socket
import subprocess

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sharded_driver_test.cpp")
HDR = os.path.join(ROOT, "cozo_amd", "csrc", "sharded_pagerank.hpp")
HDR2 = os.path.join(ROOT, "cozo_amd", "csrc", "sharded_traversal.hpp")
SO = os.path.join(ROOT, "tests", "cpp", "bin", "libsharded_driver_test.so")

AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_uint64)
AR32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_uint64)
AR64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64)


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", SRC, "-o", SO])
    return SO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, g, tol, max_iter, exchange, poison_at, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = C.CDLL(SO)
    n = g["n"]
    per = (n + world - 1) // world
    rb, re = min(n, rank * per), min(n, (rank + 1) * per)
    ioff = g["ioff"].astype(np.uint64)
    off_local = np.ascontiguousarray(ioff[rb:re + 1] - ioff[rb])
    src = np.ascontiguousarray(g["isrc"][int(ioff[rb]):int(ioff[re])], dtype=np.uint32)
    outdeg = np.ascontiguousarray(g["outdeg"], dtype=np.uint32)
    poison = np.zeros(1, dtype=np.uint8)
    calls = {"ag": 0}

    def ag(_ctx, buf, per_):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(per_) * world,)))
        dist.all_gather_into_tensor(t, t[rank * per_:(rank + 1) * per_].clone())
        calls["ag"] += 1
        if poison_at is not None and rank == poison_at[0] and calls["ag"] == poison_at[1]:
            poison[0] = 1  # this rank's Poison is set between two iterations; the other rank never sees its own flag
        return 0

    def ar32(_ctx, buf, n_):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(n_),)))
        dist.all_reduce(t)
        calls["ag"] += 1
        if poison_at is not None and rank == poison_at[0] and calls["ag"] == poison_at[1]:
            poison[0] = 1
        return 0

    def ar64(_ctx, buf, n_):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(n_),)))
        dist.all_reduce(t)
        return 0

    cbs = (AG(ag), AR32(ar32), AR64(ar64))
    scores = np.zeros(re - rb, dtype=np.float32)
    it = C.c_uint32(0)
    err = C.c_double(0)
    counters = (C.c_int * 3)()
    L.cz_test_sharded_pagerank_host.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_float, C.c_double, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, AG, AR32, AR64,
                                                C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    rc = L.cz_test_sharded_pagerank_host(n, per, rank, world, off_local.ctypes.data, src.ctypes.data, outdeg.ctypes.data,
                                         np.float32(0.85), float(tol), int(max_iter), int(exchange), poison.ctypes.data, None,
                                         *cbs, scores.ctypes.data, C.byref(it), C.byref(err), counters)
    q.put((rank, rb, re, rc, scores, it.value, err.value, list(counters)))
    dist.barrier()
    dist.destroy_process_group()


def _run(g, tol, max_iter, exchange, poison_at=None, world=2):
    build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, g, tol, max_iter, exchange, poison_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


@pytest.mark.parametrize("exchange", [0, 1], ids=["all_gather", "all_reduce"])
@pytest.mark.parametrize("n,e,tol,max_iter", [(301, 2500, 1e-4, 10), (64, 400, 0.0, 5), (7, 12, 1e-4, 10), (1000, 9000, 0.0, 20)])
def test_cpp_sharded_loop_world2_bit_identical(oracle, n, e, tol, max_iter, exchange):
    frm, to = util.random_relation(n, e, 23)
    g = util.graph_from_relation(oracle, frm, to)
    want, want_it, want_err = oracle.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, max_iter)
    out = _run(g, tol, max_iter, exchange)
    got = np.concatenate([o[4] for o in out])
    assert all(o[3] == 0 for o in out)
    assert all(o[5] == want_it for o in out), "every rank stops at the same iteration"
    assert np.array_equal(got, want), "scores bit-identical to the single-process oracle"
    assert all(o[6] == pytest.approx(want_err, rel=1e-12) for o in out)
    # one sweep, one exchange and one error reduction per iteration on every rank
    assert all(o[7] == [want_it, want_it if exchange == 0 else 0, want_it] for o in out)


def test_cpp_sharded_loop_cancellation_is_collective(oracle):
    """Poison set on ONE rank after its 3rd exchange: both ranks leave with the cancelled code at the same point (tolerance > 0:
    the reduced flag is read every iteration) instead of one of them blocking in the next collective."""
    frm, to = util.random_relation(400, 3000, 5)
    g = util.graph_from_relation(oracle, frm, to)
    out = _run(g, 1e-12, 50, 0, poison_at=(1, 3))
    assert [o[3] for o in out] == [1, 1]  # czs::RUN_CANCELLED on both
    assert out[0][7][1] == out[1][7][1] == 4  # ... after the same number of exchanges
    assert out[0][7][0] == 4 and out[1][7][0] == 3  # the poisoned rank skipped its last sweep
    # tolerance <= 0 (sweeps back to back): the flag is looked at every 8th iteration
    out = _run(g, 0.0, 50, 0, poison_at=(0, 2))
    assert [o[3] for o in out] == [1, 1] and out[0][7][1] == out[1][7][1] == 8


# ---- ONE BFS / ONE SSSP over a vertex-partitioned graph (cozo_amd/csrc/sharded_traversal.hpp) -------------------------
ARU32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint64, C.c_int)
ARU64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_int)


def _traversal_worker(rank, world, port, what, g, start, goals, poison_at, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = C.CDLL(SO)
    n = g["n"]
    per = (n + world - 1) // world
    rb, re = min(n, rank * per), min(n, (rank + 1) * per)
    ooff = g["ooff"].astype(np.uint64)
    off_local = np.ascontiguousarray(ooff[rb:re + 1] - ooff[rb])
    tgt = np.ascontiguousarray(g["otgt"][int(ooff[rb]):int(ooff[re])], dtype=np.uint32)
    poison = np.zeros(1, dtype=np.uint8)
    calls = {"n": 0}

    def _reduce(buf, n_, op, dtype):
        a = np.ctypeslib.as_array(buf, shape=(int(n_),))
        # gloo has no unsigned types: the values fit the signed ones except the all-ones "none" words, which MIN must keep
        # as the largest value -- reduce on the bit-flipped sign image
        t = torch.from_numpy((a ^ dtype(1 << (a.itemsize * 8 - 1))).view(np.int64 if a.itemsize == 8 else np.int32).copy())
        if op == 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        else:
            base = dtype(1 << (a.itemsize * 8 - 1))
            t = torch.from_numpy(a.astype(np.int64).copy())  # sums stay far below 2^63
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            a[:] = t.numpy().astype(a.dtype)
            calls["n"] += 1
            if poison_at is not None and rank == poison_at[0] and calls["n"] == poison_at[1]:
                poison[0] = 1
            return 0
        a[:] = t.numpy().view(a.dtype) ^ dtype(1 << (a.itemsize * 8 - 1))
        return 0

    ar32 = ARU32(lambda _c, buf, n_, op: _reduce(buf, n_, op, np.uint32))
    ar64 = ARU64(lambda _c, buf, n_, op: _reduce(buf, n_, op, np.uint64))
    if what == "bfs":
        parent = np.empty(n, np.uint32)
        depth = np.empty(n, np.uint32)
        order = np.empty(n, np.uint32)
        reached = C.c_uint32(0)
        gl = None if goals is None else np.ascontiguousarray(goals, dtype=np.uint32)
        L.cz_test_sharded_bfs_host.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                               C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, ARU32, ARU64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.POINTER(C.c_uint32)]
        rc = L.cz_test_sharded_bfs_host(n, rb, re, off_local.ctypes.data, tgt.ctypes.data, start,
                                        None if gl is None else gl.ctypes.data, 0 if gl is None else gl.size, int(gl is not None),
                                        poison.ctypes.data, None, ar32, ar64, parent.ctypes.data, depth.ctypes.data,
                                        order.ctypes.data, C.byref(reached))
        q.put((rank, rc, parent, depth, order[:reached.value].copy(), reached.value))
    elif what == "cc":
        group = np.empty(n, np.uint32)
        k = C.c_uint32(0)
        counters = np.zeros(2, np.uint32)
        L.cz_test_sharded_cc_host.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ARU32,
                                              C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        rc = L.cz_test_sharded_cc_host(n, rb, re, off_local.ctypes.data, tgt.ctypes.data, poison.ctypes.data, None, ar32,
                                       group.ctypes.data, C.byref(k), counters.ctypes.data)
        q.put((rank, rc, group, k.value, counters.copy()))
    else:
        w = np.ascontiguousarray(g["ow"][int(ooff[rb]):int(ooff[re])], dtype=np.float32)
        dist_out = np.empty(n, np.float32)
        parent = np.empty(n, np.uint32)
        counters = np.zeros(4, np.uint64)
        # the bucket width: the mean edge weight of the whole graph unless the test forces one (0 = one pile)
        delta = float(np.mean(g["ow"])) if g.get("delta") is None and len(g["ow"]) else float(g.get("delta") or 0.0)
        L.cz_test_sharded_sssp_host.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, ARU32, ARU64, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
        rc = L.cz_test_sharded_sssp_host(n, rb, re, rank, world, delta, off_local.ctypes.data, tgt.ctypes.data, w.ctypes.data, start,
                                         poison.ctypes.data, None, ar32, ar64, dist_out.ctypes.data, parent.ctypes.data,
                                         counters.ctypes.data)
        q.put((rank, rc, dist_out, parent, counters.copy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_traversal(what, g, start, goals=None, poison_at=None, world=2):
    build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_traversal_worker, args=(r, world, port, what, g, start, goals, poison_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out, key=lambda o: o[0])


@pytest.mark.parametrize("n,e,seed", [(400, 1500, 1), (2000, 9000, 2), (50, 60, 3)])
def test_vertex_partitioned_bfs_world2_is_the_fifo_bfs(oracle, n, e, seed):
    """parents (first discoverers), depths and the discovery order of ONE BFS whose frontier entries are expanded by the
    rank that owns them == the single-process FIFO BFS of the oracle, on both ranks"""
    frm, to = util.random_relation(n, e, seed)
    g = util.graph_from_relation(oracle, frm, to)
    want_order, want_parent, _ = oracle.bfs_order(g["n"], g["ooff"], g["otgt"], 0)
    out = _run_traversal("bfs", g, 0)
    for _, rc, parent, depth, order, reached in out:
        assert rc == 0 and reached == len(want_order)
        assert np.array_equal(order, want_order), "discovery order differs from the reference's FIFO order"
        assert np.array_equal(parent, want_parent)
        assert depth[0] == 0 and (depth[want_order] != 0xFFFFFFFF).all()
    # with goals: the traversal stops after the level in which the last goal was discovered; goal paths equal the oracle's
    goals = np.array([g["n"] - 1, g["n"] // 2, 3], dtype=np.uint32)
    wp = oracle.shortest_path_bfs(g["n"], g["ooff"], g["otgt"], 0, goals)
    out = _run_traversal("bfs", g, 0, goals=goals)
    for _, rc, parent, depth, order, reached in out:
        assert rc == 0
        for t in goals.tolist():
            assert oracle.path_from_parent(parent, 0, t) == oracle.path_from_parent(wp, 0, t)


@pytest.mark.parametrize("n,e,seed", [(400, 1500, 4), (1500, 7000, 5)])
def test_vertex_partitioned_sssp_world2_costs_are_dijkstras(oracle, n, e, seed):
    frm, to = util.random_relation(n, e, seed)
    rng = np.random.default_rng(seed)
    g = util.graph_from_relation(oracle, frm, to, weights=(rng.integers(1, 40, len(frm)) / 4).astype(np.float64))
    want_dist, _ = oracle.dijkstra(g["n"], g["ooff"], g["otgt"], g["ow"], 0)
    out = _run_traversal("sssp", g, 0)
    ooff = g["ooff"].astype(np.int64)
    for _, rc, d, parent, _counters in out:
        assert rc == 0 and np.array_equal(d, want_dist), "f32 costs must be bit-identical to Dijkstra's"
        # every parent is a tight predecessor of strictly smaller cost -- the smallest such node -- on both ranks alike
        for v in range(g["n"]):
            if v == 0 or not np.isfinite(d[v]):
                assert parent[v] == 0xFFFFFFFF
                continue
            tight = [u for u in range(g["n"]) if np.isfinite(d[u]) and d[u] < d[v] and any(
                g["otgt"][k] == v and np.float32(d[u] + g["ow"][k]) == d[v] for k in range(ooff[u], ooff[u + 1]))] if g["n"] <= 400 else None
            if tight is not None:
                assert parent[v] == min(tight)
    assert np.array_equal(out[0][3], out[1][3])


@pytest.mark.parametrize("delta,world", [(None, 2), (0.0, 2), (0.25, 2), (1e-9, 2), (1e9, 2), (None, 3), (0.25, 4)])
def test_vertex_partitioned_sssp_schedule_never_changes_the_result(oracle, delta, world):
    """the near-far schedule of the sharded loop under every bucket width (the mean weight, one pile, a narrow one, one the f32
    sum absorbs -- the threshold must still move --, one wider than every path) and over 2 and 3 ranks: costs == Dijkstra's,
    parents equal on every rank; and the exchange is SPARSE: what crossed it is a small multiple of the pairs, far below the
    rounds x N words the first form of the loop all-reduced"""
    frm, to = util.random_relation(1200, 6000, 11)
    rng = np.random.default_rng(11)
    g = util.graph_from_relation(oracle, frm, to, weights=(rng.integers(1, 40, len(frm)) / 4).astype(np.float64))
    g["delta"] = delta
    want_dist, _ = oracle.dijkstra(g["n"], g["ooff"], g["otgt"], g["ow"], 0)
    out = _run_traversal("sssp", g, 0, world=world)
    for _, rc, d, parent, counters in out:
        assert rc == 0 and np.array_equal(d, want_dist)
        assert np.array_equal(parent, out[0][3])
        rounds, exchanges, pairs, words = (int(x) for x in counters)
        assert np.array_equal(counters, out[0][4]), "every rank saw the same rounds and the same pairs"
        assert exchanges <= 2 * rounds + 1  # per round: the counts, and the pairs when there were any; + the parents at the end
        assert pairs >= np.isfinite(want_dist).sum() - 1  # every reached node but the start was proposed at least once
        assert words <= 2 * world * pairs + 2 * world * rounds  # padding to the longest list at most doubles... per rank and round
        assert words < rounds * g["n"], "the exchange must not scale with rounds x N"
    if delta is None and world == 2:  # the mean weight: more rounds than one pile, fewer relaxations (pairs)
        g1 = dict(g, delta=0.0)
        out1 = _run_traversal("sssp", g1, 0, world=2)
        assert int(out1[0][4][2]) > int(out[0][4][2]), "one pile proposes more pairs than the near-far schedule"


def test_vertex_partitioned_sssp_zero_weights_and_unreachable_nodes(oracle):
    """all-zero weights (mean 0 = one pile; every cost 0.0) and a graph whose second half is unreachable"""
    frm, to = util.random_relation(300, 900, 5)
    keep = frm < 150  # nothing leaves... the nodes >= 150 have no out-edges; some are still reached
    g = util.graph_from_relation(oracle, frm[keep], to[keep], weights=np.zeros(int(keep.sum())))
    want_dist, _ = oracle.dijkstra(g["n"], g["ooff"], g["otgt"], g["ow"], 0)
    out = _run_traversal("sssp", g, 0)
    for _, rc, d, parent, _c in out:
        assert rc == 0 and np.array_equal(d, want_dist)
    assert np.array_equal(out[0][3], out[1][3])


def test_vertex_partitioned_sssp_cancellation_is_collective(oracle):
    frm, to = util.random_relation(600, 2500, 9)
    rng = np.random.default_rng(9)
    g = util.graph_from_relation(oracle, frm, to, weights=(rng.integers(1, 40, len(frm)) / 4).astype(np.float64))
    out = _run_traversal("sssp", g, 0, poison_at=(1, 3))
    assert [o[1] for o in out] == [1, 1]  # czs::TRAVERSAL_CANCELLED on both ranks, in the same round


def test_vertex_partitioned_bfs_cancellation_is_collective(oracle):
    frm, to = util.random_relation(600, 1500, 9)
    g = util.graph_from_relation(oracle, frm, to)
    out = _run_traversal("bfs", g, 0, poison_at=(1, 4))
    assert [o[1] for o in out] == [1, 1]  # czs::TRAVERSAL_CANCELLED on both ranks, at the same level


@pytest.mark.parametrize("n,e,seed", [(500, 420, 1), (3000, 2500, 2), (2000, 9000, 3), (40, 0, 4)])
def test_vertex_partitioned_cc_world2_numbers_the_groups_like_the_rule(oracle, n, e, seed):
    """ConnectedComponents over a 2-way vertex partition of the symmetrised graph (one all-reduce(min) of the pointer vector
    per round, the loop of sharded_traversal.hpp with a host backend over gloo) == Tarjan's numbering over ascending roots (the
    oracle), on both ranks; sparse graphs with hundreds of components, a dense one, one without edges"""
    if e:
        frm, to = util.random_relation(n, e, seed)
        g = util.graph_from_relation(oracle, frm, to, undirected=True)
    else:
        g = dict(n=n, ooff=np.zeros(n + 1, dtype=np.uint64), otgt=np.zeros(0, dtype=np.uint32))
    want, want_k = oracle.tarjan_groups(g["n"], g["ooff"], g["otgt"])
    out = _run_traversal("cc", g, 0)
    for _, rc, group, k, counters in out:
        assert rc == 0 and k == want_k and np.array_equal(group, want)
        assert 1 <= counters[0] <= 8 and counters[1] == 2 * counters[0]  # per round: the cancellation word + the labels
    assert np.array_equal(out[0][4], out[1][4])


def test_vertex_partitioned_cc_cancellation_is_collective(oracle):
    frm, to = util.random_relation(3000, 2500, 7)
    g = util.graph_from_relation(oracle, frm, to, undirected=True)
    out = _run_traversal("cc", g, 0, poison_at=(1, 1))
    assert [o[1] for o in out] == [1, 1]



# ---- the overlapped form of the same loop (czs::run_sharded_pagerank_overlapped) ---------------------------------------------
XP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_uint64, C.c_uint64)


def _worker_overlapped(rank, world, port, g, tol, max_iter, half, poison_at, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = C.CDLL(SO)
    n = g["n"]
    per = (n + world - 1) // world
    rb, re = min(n, rank * per), min(n, (rank + 1) * per)
    ioff = g["ioff"].astype(np.uint64)
    off_local = np.ascontiguousarray(ioff[rb:re + 1] - ioff[rb])
    src = np.ascontiguousarray(g["isrc"][int(ioff[rb]):int(ioff[re])], dtype=np.uint32)
    outdeg = np.ascontiguousarray(g["outdeg"], dtype=np.uint32)
    poison = np.zeros(1, dtype=np.uint8)
    calls = {"xp": 0}

    def xp(_ctx, buf, per_, lo, cnt):
        # every rank's piece [lo, lo + cnt) of its slice lands at buf + r * per + lo: a list all-gather into views
        full = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(per_) * world,)))
        views = [full[r * per_ + lo:r * per_ + lo + cnt] for r in range(world)]
        outs = [torch.empty(int(cnt), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(outs, views[rank].clone())
        for r in range(world):
            views[r].copy_(outs[r])
        calls["xp"] += 1
        if poison_at is not None and rank == poison_at[0] and calls["xp"] == poison_at[1]:
            poison[0] = 1
        return 0

    def ar64(_ctx, buf, n_):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(n_),)))
        dist.all_reduce(t)
        return 0

    cbs = (XP(xp), AR64(ar64))
    scores = np.zeros(re - rb, dtype=np.float32)
    it = C.c_uint32(0)
    err = C.c_double(0)
    counters = (C.c_int * 3)()
    L.cz_test_sharded_pagerank_overlapped_host.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                           C.c_void_p, C.c_float, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, XP, AR64,
                                                           C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    rc = L.cz_test_sharded_pagerank_overlapped_host(n, per, half, rank, world, off_local.ctypes.data, src.ctypes.data, outdeg.ctypes.data,
                                                    np.float32(0.85), float(tol), int(max_iter), poison.ctypes.data, None, *cbs,
                                                    scores.ctypes.data, C.byref(it), C.byref(err), counters)
    q.put((rank, rb, re, rc, scores, it.value, err.value, list(counters)))
    dist.barrier()
    dist.destroy_process_group()


def _run_overlapped(g, tol, max_iter, half, poison_at=None, world=2):
    build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlapped, args=(r, world, port, g, tol, max_iter, half, poison_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


@pytest.mark.parametrize("n,e,tol,max_iter,half_of", [(301, 2500, 1e-4, 10, 0.5), (64, 400, 0.0, 5, 0.25), (7, 12, 1e-4, 10, 0.5),
                                                      (1000, 9000, 0.0, 12, 0.0), (1000, 9000, 0.0, 12, 1.0)])
def test_cpp_overlapped_loop_world2_bit_identical(oracle, n, e, tol, max_iter, half_of):
    """run_sharded_pagerank_overlapped (what cz_pagerank_sharded_overlapped / cz_pagerank_multi with CZ_PR_OVERLAP_EXCHANGE run over
    RCCL): the rank's rows in two parts, the exchange of part 0 begun before part 1 is swept.  Scores, iteration count and error
    equal the single-process oracle's whatever the cut -- including the degenerate cuts (everything in one part)."""
    frm, to = util.random_relation(n, e, 23)
    g = util.graph_from_relation(oracle, frm, to)
    per = (g["n"] + 1) // 2
    half = int(per * half_of)
    want, want_it, want_err = oracle.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, max_iter)
    out = _run_overlapped(g, tol, max_iter, half)
    got = np.concatenate([o[4] for o in out])
    assert all(o[3] == 0 for o in out)
    assert all(o[5] == want_it for o in out)
    assert np.array_equal(got, want), "scores bit-identical to the single-process oracle"
    assert all(o[6] == pytest.approx(want_err, rel=1e-12) for o in out)
    assert all(o[7] == [2 * want_it, 2 * want_it, want_it] for o in out)  # two part sweeps, two exchanges begun, one reduction per iteration


def test_cpp_overlapped_loop_cancellation_is_collective(oracle):
    frm, to = util.random_relation(400, 3000, 5)
    g = util.graph_from_relation(oracle, frm, to)
    out = _run_overlapped(g, 1e-12, 50, 100, poison_at=(1, 5))
    assert all(o[3] == 1 for o in out), "both ranks return RUN_CANCELLED"
    assert out[0][7][1:] == out[1][7][1:], "and leave at the same iteration (same exchanges, same reductions)"
    assert out[1][7][0] == out[0][7][0] - 2, "the poisoned rank skipped the two part sweeps of its last iteration"

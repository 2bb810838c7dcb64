"""Child process of tests/test_gpu_comm.py: the multi-GPU entry points of the C ABI on ONE GPU (RCCL communicators of one
rank).  Runs in its own process and leaves with os._exit: a process in which BOTH PyTorch and libcozo_gpu have used the
RCCL shared library was observed to die in the library's static destructors at interpreter exit ("double free or
corruption", after every check had passed) -- scratch/r2_rccl_exit.py narrows it down.  PyTorch is imported FIRST (the
order bench.py uses), every buffer handed to a collective is owned by libcozo_gpu."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    from cozo_amd import _lib
    L = _lib.lib()
    import torch
    assert L.cz_init(0) == 0, L.cz_last_error()
    from cozo_amd import graph as G
    from cozo_amd.comm import Comm, pagerank_multi
    from cozo_amd.hnsw import HnswSearch
    from oracle import oracle as O
    from tests import util
    O.build()
    comm = Comm(Comm.unique_id(), 0, 1)
    assert comm.rank == 0 and comm.world == 1

    frm, to = util.random_relation(30000, 200000, 3)
    g = util.graph_from_relation(O, frm, to)
    for allreduce in (False, True):
        for tol, iters in ((1e-4, 10), (0.0, 20)):
            want, want_it, want_err = O.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, iters)
            plan = G.PageRankPlan(g["ioff"], g["isrc"], g["outdeg"], g["n"], 0, g["n"], 0.85)
            it, err = comm.pagerank_sharded(plan, g["n"], tol, iters, allreduce_exchange=allreduce)
            assert it == want_it and np.array_equal(plan.read_scores(), want) and abs(err - want_err) <= 1e-9 * abs(want_err)
            try:  # cancellation through the collective path
                comm.pagerank_sharded(plan, g["n"], tol, iters, poison=np.ones(1, dtype=np.uint8))
                raise AssertionError("a set poison flag must cancel the run")
            except _lib.ProcessKilled:
                pass
            # the bare exchange steps on a library-owned device buffer (the plan's scores): one rank = the identity
            ptr = plan.scores_ptr()
            before = plan.read_scores()
            comm.all_gather(ptr, g["n"] * 4)
            comm.all_reduce_sum_f64(ptr, g["n"] // 2)
            torch.cuda.synchronize()
            assert np.array_equal(plan.read_scores(), before)
            plan.close()
            s, it2, _ = pagerank_multi(g["ioff"], g["isrc"], g["outdeg"], 1, 0.85, tol, iters, allreduce_exchange=allreduce)
            assert it2 == want_it and np.array_equal(s, want)
    # the overlapped exchange: the rank's rows as two plans, one rank (the exchanges are the identity, the order of work is not)
    for tol, iters in ((1e-4, 10), (0.0, 20)):
        want, want_it, want_err = O.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, iters)
        n, mid = g["n"], g["n"] // 3
        ioff = g["ioff"].astype(np.int64)
        pa = G.PageRankPlan((ioff[:mid + 1] - ioff[0]).astype(np.uint32), g["isrc"][:ioff[mid]], g["outdeg"], n, 0, mid, 0.85)
        pb = G.PageRankPlan((ioff[mid:] - ioff[mid]).astype(np.uint32), g["isrc"][ioff[mid]:], g["outdeg"], n, mid, n, 0.85)
        it, err = comm.pagerank_sharded_overlapped(pa, pb, n, mid, tol, iters)
        got = np.concatenate([pa.read_scores(), pb.read_scores()])
        assert it == want_it and np.array_equal(got, want) and abs(err - want_err) <= 1e-9 * abs(want_err)
        pa.close()
        pb.close()
        s, it2, _ = pagerank_multi(g["ioff"], g["isrc"], g["outdeg"], 1, 0.85, tol, iters, overlap_exchange=True)
        assert it2 == want_it and np.array_equal(s, want)
    print("OK pagerank_sharded / pagerank_sharded_overlapped / pagerank_multi / collectives", flush=True)

    x = util.vectors(3000, 96, 42, "lowrank")
    _, flat = util.build_index(O, x, 1, 12, 60)
    gix = util.gpu_index(flat, "Cosine", 12)
    q = util.vectors(50, 96, 43, "lowrank")
    ids, dist, cnt = gix.hnsw_knn_batch(q, HnswSearch(k=10, ef=64))
    dev = torch.device("cuda:0")
    qd = torch.from_numpy(q).to(dev)
    oi = torch.empty((50, 10), dtype=torch.int64, device=dev)
    od = torch.empty((50, 10), dtype=torch.float64, device=dev)
    oc = torch.empty(50, dtype=torch.int32, device=dev)
    comm.hnsw_search_sharded(gix, qd, 50, 10, 64, 1000, oi, od, oc)
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy(), ids.astype(np.int64) + 1000)
    assert np.array_equal(od.cpu().numpy(), dist) and np.array_equal(oc.cpu().numpy().astype(np.uint32), cnt)
    gix.close()
    print("OK hnsw_search_sharded", flush=True)
    # the same partitioned index held by one process (cz_hnsw_multi_*): with one device, one shard == the plain index
    from cozo_amd.comm import HnswMulti
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    man = HnswIndexManifest(vec_dim=96, distance="Cosine", m_neighbours=12, ef_construction=60)
    multi = HnswMulti.build(man, x, 1, seed=5, max_batch=64)
    plain = GpuHnswIndex.build(man, x, seed=5, max_batch=64)
    mi, md, mc = multi.search(q, 10, 64)
    pi, pd, pc = plain.hnsw_knn_batch(q, HnswSearch(k=10, ef=64))
    assert multi.n_gpus == 1 and multi.id_offsets().tolist() == [0] and multi.build_n_dist > 0
    assert np.array_equal(mi, pi.astype(np.uint64)) and np.array_equal(md, pd) and np.array_equal(mc, pc)
    try:
        HnswMulti.build(man, x, 64)
        raise AssertionError("more GPUs than the box has must be refused")
    except _lib.CozoGpuError:
        pass
    multi.close()
    plain.close()
    print("OK hnsw_multi", flush=True)
    # ONE traversal over a "vertex-partitioned" graph of one rank == the single-GPU rule on the same graph
    from cozo_amd.comm import bfs_sharded, sssp_sharded, sssp_sharded_last_stats
    frm, to = util.random_relation(20000, 90000, 8)
    rng = np.random.default_rng(8)
    gw = util.graph_from_relation(O, frm, to, weights=(rng.integers(1, 40, len(frm)) / 4).astype(np.float64))
    starts = np.array([0, 7, 123], dtype=np.uint32)
    goals = np.array([19999, 5000, 42], dtype=np.uint32)
    for gl in (None, goals):
        a = G.bfs(gw["ooff"], gw["otgt"], starts, goals=gl, want_depth=True, want_order=True)
        b = bfs_sharded(comm, gw["ooff"], gw["otgt"], gw["n"], 0, gw["n"], starts, goals=gl, want_depth=True, want_order=True)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), "cz_bfs_sharded differs from cz_bfs"
    a = G.bfs(gw["ooff"], gw["otgt"], starts, share_visited=True, want_order=True)
    b = bfs_sharded(comm, gw["ooff"], gw["otgt"], gw["n"], 0, gw["n"], starts, share_visited=True, want_order=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    d0, p0 = G.sssp(gw["ooff"], gw["otgt"], gw["ow"], starts)
    d1, p1 = sssp_sharded(comm, gw["ooff"], gw["otgt"], gw["ow"], gw["n"], 0, gw["n"], starts)
    assert np.array_equal(d0, d1) and np.array_equal(p0, p1), "cz_sssp_sharded differs from cz_sssp"
    want, _ = O.dijkstra(gw["n"], gw["ooff"], gw["otgt"], gw["ow"], 7)
    assert np.array_equal(d1[1], want)
    # the near-far schedule of the sharded loop never changes a result: every bucket width (one pile, narrow, one the f32 sum
    # absorbs, wider than any path), and a far pile whose stale entries are dropped again and again
    for env in ({"CZ_SSSP_DELTA": "0"}, {"CZ_SSSP_DELTA": "2.0", "CZ_SSSP_FAR_COMPACT": "64"}, {"CZ_SSSP_DELTA": "1e-9"}, {"CZ_SSSP_DELTA": "1e9"}):
        os.environ.update(env)
        try:
            dx, px = sssp_sharded(comm, gw["ooff"], gw["otgt"], gw["ow"], gw["n"], 0, gw["n"], starts)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert np.array_equal(d0, dx) and np.array_equal(p0, px), f"cz_sssp_sharded under {env} differs from cz_sssp"
        st = sssp_sharded_last_stats()
        assert st["rounds"] > 0 and st["pairs"] >= 3 * (np.isfinite(d0[0]).sum() - 1) // 2, st
        if "CZ_SSSP_FAR_COMPACT" in env:
            assert st["compactions"] > 0 and st["buckets"] > 3, f"the far pile was never compacted: {st}"
        if env["CZ_SSSP_DELTA"] in ("0", "1e9"):
            assert st["buckets"] == 0, st  # one pile / everything below the first threshold
    # a rank that owns only part of the rows relaxes only those: the SSSP of the graph without the other rows' edges
    half = gw["n"] // 2
    o64 = gw["ooff"].astype(np.int64)
    for rb_, re_ in ((0, half), (half, gw["n"])):
        sub_off = np.zeros(gw["n"] + 1, dtype=np.int64)
        deg = np.diff(o64)
        deg[:rb_] = 0
        deg[re_:] = 0
        sub_off[1:] = np.cumsum(deg)
        sub_tgt = gw["otgt"][o64[rb_]:o64[re_]]
        sub_w = gw["ow"][o64[rb_]:o64[re_]]
        dw, pw = G.sssp(sub_off.astype(np.uint32), sub_tgt, sub_w, starts)
        ds, ps = sssp_sharded(comm, (o64[rb_:re_ + 1] - o64[rb_]).astype(np.uint32), sub_tgt, sub_w, gw["n"], rb_, re_, starts)
        assert np.array_equal(dw, ds) and np.array_equal(pw, ps), f"cz_sssp_sharded over rows [{rb_}, {re_}) differs"
    print("OK bfs_sharded / sssp_sharded", flush=True)
    from cozo_amd.comm import connected_components_sharded
    fr2, to2 = util.random_relation(30000, 28000, 9)  # sparse: hundreds of components
    gu = util.graph_from_relation(O, fr2, to2, undirected=True)
    g0, k0 = G.connected_components(gu["ooff"], gu["otgt"])
    g1, k1, rounds = connected_components_sharded(comm, gu["ooff"], gu["otgt"], gu["n"], 0, gu["n"])
    assert k0 == k1 and k0 > 100 and np.array_equal(g0, g1) and rounds >= 1, "cz_connected_components_sharded differs"
    print(f"OK connected_components_sharded ({k1} components, {rounds} rounds)", flush=True)
    # the single-process forms (one host thread + one communicator per device; one device here): the one-GPU rules' results
    from cozo_amd.comm import bfs_multi, sssp_multi, connected_components_multi
    a = G.bfs(gw["ooff"], gw["otgt"], starts, goals=goals, want_depth=True, want_order=True)
    b = bfs_multi(gw["ooff"], gw["otgt"], 1, starts, goals=goals, want_depth=True, want_order=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)), "cz_bfs_multi differs from cz_bfs"
    d2, p2 = sssp_multi(gw["ooff"], gw["otgt"], gw["ow"], 1, starts)
    assert np.array_equal(d0, d2) and np.array_equal(p0, p2), "cz_sssp_multi differs from cz_sssp"
    g2, k2 = connected_components_multi(gu["ooff"], gu["otgt"], 1)
    assert k2 == k0 and np.array_equal(g2, g0), "cz_connected_components_multi differs"
    print("OK bfs_multi / sssp_multi / connected_components_multi", flush=True)
    # a shard whose targets point outside the graph, or whose offsets go backwards, is refused on the host arrays (the kernels
    # index full-length per-node arrays by the targets)
    bad_t = gw["otgt"].copy()
    bad_t[5] = gw["n"] + 7
    bad_o = gw["ooff"].copy()
    row = int(np.flatnonzero(np.diff(bad_o.astype(np.int64)) > 0)[3])  # a row with edges: its end pulled below its start
    bad_o[row + 1] = bad_o[row] - 1 if bad_o[row] > 0 else 0
    if bad_o[row + 1] >= bad_o[row]:
        bad_o[row] += 2
    for off_, tgt_ in ((gw["ooff"], bad_t), (bad_o, gw["otgt"])):
        if np.array_equal(off_, gw["ooff"]) and np.array_equal(tgt_, gw["otgt"]):
            continue
        for call in (lambda: bfs_sharded(comm, off_, tgt_, gw["n"], 0, gw["n"], starts),
                     lambda: sssp_sharded(comm, off_, tgt_, gw["ow"], gw["n"], 0, gw["n"], starts),
                     lambda: connected_components_sharded(comm, off_, tgt_, gw["n"], 0, gw["n"])):
            try:
                call()
                raise AssertionError("a malformed shard must be refused")
            except _lib.CozoGpuError as e:
                assert "out of range" in str(e) or "monotone" in str(e), str(e)
    print("OK malformed shards refused", flush=True)
    comm.close()
    print("ALL OK", flush=True)


if __name__ == "__main__":
    code = 1
    try:
        main()
        code = 0
    except BaseException as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)

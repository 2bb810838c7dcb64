"""The multi-GPU entry points of the C ABI on ONE GPU: RCCL communicators of one rank (cz_comm_create_rank with world = 1,
cz_pagerank_multi with n_gpus = 1).  The collectives degenerate to copies, but everything else is the code that runs at
N > 1: RCCL is loaded and initialised, the sharded loop (cozo_amd/csrc/sharded_pagerank.hpp) runs over its HIP backend, the
sharded search packs / gathers / merges.  The exchange logic itself is covered with world_size 2 in tests/test_sharded_driver.py."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(gpu_lib):
    from cozo_amd.comm import Comm
    c = Comm(Comm.unique_id(), 0, 1)
    yield c
    c.close()


def test_collectives_of_one_rank(comm):
    import torch
    dev = torch.device("cuda:0")
    x = torch.arange(1000, dtype=torch.float32, device=dev)
    comm.all_gather(x, 4000)
    e = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev)
    comm.all_reduce_sum_f64(e, 2)
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32)) and e.tolist() == [1.5, 2.0]


@pytest.mark.parametrize("allreduce", [False, True])
@pytest.mark.parametrize("tol,iters", [(1e-4, 10), (0.0, 20)])
def test_pagerank_sharded_and_multi_match_oracle(comm, oracle, tol, iters, allreduce):
    from cozo_amd import graph as G
    from cozo_amd.comm import pagerank_multi
    frm, to = util.random_relation(30000, 200000, 3)
    g = util.graph_from_relation(oracle, frm, to)
    want, want_it, want_err = oracle.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, tol, iters)
    plan = G.PageRankPlan(g["ioff"], g["isrc"], g["outdeg"], g["n"], 0, g["n"], 0.85)
    it, err = comm.pagerank_sharded(plan, g["n"], tol, iters, allreduce_exchange=allreduce)
    assert it == want_it and np.array_equal(plan.read_scores(), want) and err == pytest.approx(want_err, rel=1e-9)
    # cancellation through the collective path
    from cozo_amd import _lib
    with pytest.raises(_lib.ProcessKilled):
        comm.pagerank_sharded(plan, g["n"], tol, iters, poison=np.ones(1, dtype=np.uint8))
    plan.close()
    s, it2, _ = pagerank_multi(g["ioff"], g["isrc"], g["outdeg"], 1, 0.85, tol, iters, allreduce_exchange=allreduce)
    assert it2 == want_it and np.array_equal(s, want)


def test_hnsw_search_sharded_one_shard(comm, oracle):
    import torch
    from cozo_amd.hnsw import HnswSearch
    x = util.vectors(3000, 96, 42, "lowrank")
    _, flat = util.build_index(oracle, x, 1, 12, 60)
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

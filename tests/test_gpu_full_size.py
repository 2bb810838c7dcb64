"""BASELINE.json's full sizes on the GPU: configs[1] HNSW k=10 cosine over 1M x 768 with a 1024-query batch, configs[2]
PageRank on 10M nodes / 100M edges.  Inputs are generated on the device exactly as bench.py does.  Checked two ways:
size-independent properties (ordering, determinism, recall, formulations agreeing), and -- on a bounded sample the CPU
oracle finishes in seconds -- directly against the oracle: the GPU-built 1M index is exported and 256 of the queries are
searched by the oracle (ids, distances, evaluation counts bit-equal in the kernel's summation order, within 1e-5 relative in
the reference's), and 3 PageRank sweeps over the full 10M / 100M graph are compared score by score.  (bench.py repeats the
same oracle comparison on the 10M x 768 index of the headline metric on every run: `parity` in its JSON line.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(gpu_lib):
    import torch
    assert torch.cuda.is_available()
    return torch


def test_hnsw_1m_x_768_batch_1024_properties(torch_gpu):
    torch = torch_gpu
    import bench as Bn
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch, distance_batch_device
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    n, dim, B, k, ef = 1_000_000, 768, 1024, 10, 96
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    try:
        ids = torch.empty((B, k), dtype=torch.int32, device=dev)
        dd = torch.empty((B, k), dtype=torch.float64, device=dev)
        cnt = torch.empty(B, dtype=torch.int32, device=dev)
        nd = torch.zeros(B, dtype=torch.int64, device=dev)
        ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        torch.cuda.synchronize()
        ids64 = ids.to(torch.int64) & 0xFFFFFFFF
        # every query returns k rows, ids in range and distinct, distances ascending (hnsw.rs:1005-1006) and in [0, 2]
        assert bool((cnt == k).all()) and bool((ids64 < n).all())
        assert bool((torch.sort(ids64, dim=1).values.diff(dim=1) != 0).all())
        assert bool((dd.diff(dim=1) >= 0).all()) and float(dd.min()) >= -1e-6 and float(dd.max()) <= 2.0 + 1e-6
        assert int(nd.min()) >= ef  # at least ef distance evaluations per query
        # determinism / idempotence: the same launch twice gives the same bits
        ids2, dd2, nd2 = torch.empty_like(ids), torch.empty_like(dd), torch.zeros_like(nd)
        ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids2, dd2, cnt, nd2, stream)
        torch.cuda.synchronize()
        assert torch.equal(ids, ids2) and torch.equal(dd, dd2) and torch.equal(nd, nd2)
        # the distances the search reports are the batched-distance kernel's distances for the same (query, node) pairs
        pairs = torch.stack([torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(k), ids.reshape(-1)], 1).contiguous()
        chk = torch.empty(B * k, dtype=torch.float64, device=dev)
        distance_batch_device("Cosine", x, q, pairs, chk, stream)
        torch.cuda.synchronize()
        assert torch.equal(chk.view(B, k), dd)
        # recall against the exhaustive scan (the bench's bar is 0.95 at this ef)
        gt = torch.empty((B, k), dtype=torch.int32, device=dev)
        gtd = torch.empty((B, k), dtype=torch.float64, device=dev)
        ix.bruteforce_knn_device(q, k, gt, gtd, stream)
        torch.cuda.synchronize()
        assert Bn.recall_at_k(torch, ids64, gt.to(torch.int64) & 0xFFFFFFFF) >= 0.95
        assert bool((gtd[:, :1] <= dd[:, :1]).all())  # the exact nearest neighbour is never farther than the found one
        # a larger ef can only bring the k-th result closer
        ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=2 * ef), ids2, dd2, cnt, nd2, stream)
        torch.cuda.synchronize()
        assert float((dd2[:, -1] <= dd[:, -1]).double().mean()) >= 0.99
        # ---- against the oracle: the same index (exported), 256 of the same queries ----
        from oracle import oracle as O
        O.build()
        nodes, nbrs, entry = ix.export()
        vec = ix.export_vectors()
        flat = O.FlatIndex(vec, O.COSINE, nodes, nbrs, entry)
        nq = 256
        qh = q[:nq].cpu().numpy()
        oids, odist, ocnt, ond = flat.knn_batch(qh, k, ef, dot_mode=O.DOT_GPU, threads=32)
        gids = (ids[:nq].cpu().numpy().astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
        assert np.array_equal(gids, oids) and np.array_equal(dd[:nq].cpu().numpy(), odist)
        assert np.array_equal(cnt[:nq].cpu().numpy().astype(np.uint32), ocnt) and int(nd[:nq].sum().item()) == ond
        # the reference's own summation order (ndarray's 8-accumulator dot): same rows, distances within 1e-5 RELATIVE
        rids, rdist, _, _ = flat.knn_batch(qh, k, ef, dot_mode=O.DOT_NDARRAY, threads=32)
        same = (gids == rids).all(axis=1)
        assert same.mean() >= 0.99
        rel = np.abs(dd[:nq].cpu().numpy()[same] - rdist[same]) / np.abs(rdist[same])
        assert rel.max() <= 1e-5, rel.max()
    finally:
        ix.close()


def test_pagerank_10m_100m_formulations_agree(torch_gpu):
    """The blocked two-phase sweep and the CSR-stream gather sweep add every row's contributions in the same order:
    scores and the f64 error must be bit-identical on the full-size graph; plus the fixed-point property of one more
    sweep (the error is non-increasing over these iterations on a graph without sinks in its in-adjacency)."""
    torch = torch_gpu
    from cozo_amd.graph import PageRankPlan
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    n, e = 10_000_000, 100_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
    src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
    keep = src != dst
    key = torch.unique(dst[keep] * n + src[keep])
    del dst, src, keep
    d = torch.div(key, n, rounding_mode="floor")
    s = (key - d * n).to(torch.int32)
    del key
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(torch.bincount(d, minlength=n), 0)
    outdeg = torch.bincount(s.to(torch.int64), minlength=n).to(torch.int32)
    off32 = off.to(torch.int32)
    del d, off
    results = {}
    for mode in ("blocked", "gather"):
        plan = PageRankPlan(off32, s, outdeg, n, 0, n, 0.85, device_ptrs=True, mode=mode)
        assert plan.blocked == (mode == "blocked")
        c0 = torch.empty(n, dtype=torch.float32, device=dev)
        c1 = torch.empty_like(c0)
        errs = []
        plan.init(c0, stream)
        for _ in range(4):
            err = torch.zeros(1, dtype=torch.float64, device=dev)
            plan.step(c0, c1, err, stream)
            c0, c1 = c1, c0
            errs.append(float(err.item()))
        sc = torch.empty(n, dtype=torch.float32, device=dev)
        plan.read_scores(sc)
        torch.cuda.synchronize()
        results[mode] = (sc, errs)
        if mode == "blocked":  # ---- against the oracle: 3 sweeps over the whole graph, score by score ----
            from oracle import oracle as O
            O.build()
            h_off = off32.cpu().numpy().astype(np.uint64)
            h_src = s.cpu().numpy().astype(np.uint32)
            h_od = outdeg.cpu().numpy().astype(np.uint32)
            want, _, want_err = O.pagerank(n, h_off, h_src, h_od, 0.85, 0.0, 3, threads=64)
            plan.init(c0, stream)
            for _ in range(3):
                err = torch.zeros(1, dtype=torch.float64, device=dev)
                plan.step(c0, c1, err, stream)
                c0, c1 = c1, c0
            assert np.array_equal(plan.read_scores(), want), "10M / 100M: scores after 3 sweeps differ from the oracle"
            assert float(err.item()) == pytest.approx(want_err, rel=1e-9)
        plan.close()
        del c0, c1
    # ---- the in-place reading on the same full-size graph (round 6: the resident, level-scheduled plan): 3 sweeps == the oracle's
    # in-place mode score by score; init + sweeps reproduces run; and it is NOT the Jacobi vector
    from cozo_amd.graph import InplacePageRankPlan
    from oracle import oracle as O
    ip = InplacePageRankPlan(off32, s, outdeg, 0.85, device_ptrs=True)
    info = ip.info
    assert info["levels"] > 20 and info["graph_replay"] == 1
    assert info["x_edges"] + info["y_edges"] + info["urgent_edges"] + info["long_row_edges"] == int(s.numel())
    it, err = ip.run(0.0, 3)
    want_ip, _, want_ip_err = O.pagerank_mode(n, h_off, h_src, h_od, 0.85, 0.0, 3, mode=O.PR_INPLACE)
    got_ip = ip.read_scores()
    assert it == 3 and np.array_equal(got_ip, want_ip), "10M / 100M: in-place scores after 3 sweeps differ from the oracle"
    assert err == pytest.approx(want_ip_err, rel=1e-9)
    ip.init(stream)
    ip.sweeps(3, stream)
    assert np.array_equal(ip.read_scores(), want_ip)
    assert not np.array_equal(want_ip, want)
    ip.close()
    assert torch.equal(results["blocked"][0], results["gather"][0])
    assert results["blocked"][1] == results["gather"][1]
    errs = results["blocked"][1]
    assert all(b <= a for a, b in zip(errs, errs[1:])) and errs[-1] > 0.0
    sc = results["blocked"][0]
    assert bool(torch.isfinite(sc).all()) and float(sc.min()) >= (1 - 0.85) / n * 0.999

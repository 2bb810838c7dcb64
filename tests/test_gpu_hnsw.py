"""GPU parity: distances and HNSW k-NN search through the C ABI vs the CPU oracle (same seeded inputs).

Two different statements are made here; they are not the same strength and are kept apart:

 * BIT-EQUALITY with ORC_DOT_GPU -- an oracle mode written to restate the kernels' OWN summation tree.  It proves the CPU restatement
   of the kernel and the kernel agree (and pins the traversal: ids, visit counts).  It says nothing about the reference's arithmetic.
 * The INDEPENDENT statement: distances against ORC_DOT_NDARRAY, the reference's ndarray summation order (runtime/hnsw.rs:66-109).
   north_star's bar is 1e-5 RELATIVE to the reference's value, and that is what is asserted for every pair whose value is not the
   difference of much larger terms.  Where the value cancels (`1 - x` near 0 for Cosine / IP, L2 of nearly equal vectors: |ref| below
   5 % of the magnitude the f32 sums work at) two f32 summation orders of the same dot cannot agree tighter than eps * sum|a_i b_i|
   in ABSOLUTE terms; there the FALLBACK bound is 1e-5 of that magnitude.  The fallback is weaker than north_star's bar; the tests
   print how many pairs took it (`fallback_pairs`), and bench.py's `parity` object carries the same count for the timed batch."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
RTOL = 1e-5

METRICS = [("L2", 0), ("Cosine", 1), ("IP", 2)]


@pytest.mark.parametrize("dim", [1, 2, 7, 8, 9, 33, 100, 128, 129, 768, 1000, 1536, 2052, 2500])
@pytest.mark.parametrize("name,metric", METRICS)
def test_distance_batch_matches_oracle(gpu_lib, oracle, dim, name, metric):
    from cozo_amd.hnsw import distance_batch
    rng = np.random.default_rng(dim * 7 + metric)
    base = util.vectors(300, dim, 1 + dim, "normal")
    q = util.vectors(17, dim, 2 + dim, "normal")
    pairs = np.stack([rng.integers(0, 17, 2000), rng.integers(0, 300, 2000)], 1).astype(np.uint32)
    got = distance_batch(name, base, q, pairs)
    exact = oracle.distance_pairs(metric, base, q, pairs, oracle.DOT_GPU)
    ref = oracle.distance_pairs(metric, base, q, pairs, oracle.DOT_NDARRAY)
    assert np.array_equal(got, exact), "kernel summation tree differs from its CPU restatement"
    # north_star's bar: 1e-5 RELATIVE to the reference's value (ndarray summation order), asserted as such wherever the
    # value is not the difference of much larger terms.  `1 - x` (Cosine, IP) and L2 of nearly equal vectors cancel: two
    # f32 summation orders of the same dot cannot agree tighter than eps * sum|a_i b_i| in ABSOLUTE terms, whatever is
    # subtracted afterwards -- there (|ref| below 5 % of the magnitude the f32 sums work at) the bound is 1e-5 of that
    # magnitude instead.  Both bounds are stated here, neither folds into the other; bench.py reports the measured
    # true-relative error of the search's distances (`parity.max_rel_err_vs_reference_arithmetic`, ~1e-6).
    a64, b64 = q[pairs[:, 0]].astype(np.float64), base[pairs[:, 1]].astype(np.float64)
    mag = {0: np.sum((a64 - b64) ** 2, axis=1), 1: np.ones(len(ref)), 2: 1.0 + np.sum(np.abs(a64 * b64), axis=1)}[metric]
    finite = np.isfinite(ref)
    true_rel = np.abs(got - ref)[finite] / np.maximum(np.abs(ref[finite]), 1e-300)
    well = np.abs(ref[finite]) >= 0.05 * mag[finite]
    print(f"dim {dim} {name}: north_star bound on {int(well.sum())} pairs, fallback_pairs {int((~well).sum())}")
    if well.any():
        assert true_rel[well].max() <= RTOL, true_rel[well].max()  # north_star's bar
    assert np.max(np.abs(got - ref)[finite] / np.maximum(mag[finite], 1e-300)) <= RTOL  # the cancellation fallback (all pairs satisfy it)
    if dim in (33, 768):  # the same pairs against the vectors of an INDEX (cz_hnsw_index_distance_batch): the same bits
        _, flat = util.build_index(oracle, base, metric, 8, 32)
        gix = util.gpu_index(flat, name, 8)
        assert np.array_equal(gix.distance_batch(q, pairs), got)


@pytest.mark.parametrize("dim,nq", [(33, 17), (128, 4000), (768, 60), (1000, 300), (2052, 9)])
@pytest.mark.parametrize("name,metric", METRICS)
def test_distance_batch_grouped_by_query_path(gpu_lib, oracle, monkeypatch, dim, nq, name, metric):
    """P >= 65536 over few queries, CZ_PAIRS_GROUPED=1: cz_distance_batch groups the pairs by query (counting sort) and
    keeps the query in registers.  Same bits as the default ungrouped kernel and as the oracle, rows written back to the
    callers' positions."""
    from cozo_amd.hnsw import distance_batch
    rng = np.random.default_rng(dim + metric)
    n, P = 500, 70000
    base = util.vectors(n, dim, 3 + dim, "normal")
    q = util.vectors(nq, dim, 4 + dim, "normal")
    pairs = np.stack([rng.integers(0, nq, P), rng.integers(0, n, P)], 1).astype(np.uint32)
    pairs[:5, 0] = nq - 1  # the largest query id sits at the front of the caller's order
    monkeypatch.setenv("CZ_PAIRS_GROUPED", "1")
    grouped = distance_batch(name, base, q, pairs)
    monkeypatch.setenv("CZ_PAIRS_GROUPED", "0")
    plain = distance_batch(name, base, q, pairs)
    assert np.array_equal(grouped, plain)
    sample = rng.choice(P, 3000, replace=False)
    assert np.array_equal(grouped[sample], oracle.distance_pairs(metric, base, q, pairs[sample], oracle.DOT_GPU))


def test_distance_known_answers(gpu_lib, oracle):
    """hand-computable values mirroring runtime/tests.rs:691-697 (l2_dist / cos_dist / ip_dist)."""
    from cozo_amd.hnsw import distance_batch
    base = np.array([[2, 3], [1, 2], [0.6, 0.8]], dtype=np.float32)
    q = np.array([[1, 2], [0.6, 0.8]], dtype=np.float32)
    assert distance_batch("L2", base, q, np.array([[0, 0]], np.uint32))[0] == 2.0
    assert distance_batch("Cosine", base, q, np.array([[0, 1]], np.uint32))[0] == pytest.approx(0.0, abs=1e-7)
    assert distance_batch("IP", base, q, np.array([[1, 2]], np.uint32))[0] == pytest.approx(0.0, abs=1e-7)
    zero = distance_batch("Cosine", np.zeros((1, 2), np.float32), q, np.array([[0, 0]], np.uint32))[0]
    assert np.isnan(zero)  # zero vector under cosine -> NaN, as in the reference


CASES = [
    # n, dim, distance, metric, m, ef_c, kind
    (3000, 128, "L2", 0, 16, 100, "uniform"),
    (2000, 768, "Cosine", 1, 16, 64, "lowrank"),
    (2500, 100, "IP", 2, 8, 50, "normal"),
    (1500, 36, "L2", 0, 12, 40, "uniform"),
    (1200, 1536, "Cosine", 1, 8, 40, "lowrank"),
]


@pytest.fixture(scope="module", params=CASES, ids=lambda c: f"n{c[0]}_d{c[1]}_{c[2]}")
def case(request, oracle, gpu_lib):
    n, dim, dist, metric, m, efc, kind = request.param
    x = util.vectors(n, dim, 42, kind)
    builder, flat = util.build_index(oracle, x, metric, m, efc)
    gix = util.gpu_index(flat, dist, m)
    q = util.vectors(64, dim, 43, kind)
    yield dict(x=x, flat=flat, gix=gix, q=q, metric=metric, m=m, dist=dist)
    gix.close()


# (ef > 512: the ranked merge; ef > 1 024 and k > 1 024: beyond round 3's cap -- the list outgrows the index on the small cases,
#  i.e. every node ends up in it, and on the 3 000-node case ef = 2 500 evicts)
@pytest.mark.parametrize("ef,k", [(1, 1), (10, 10), (64, 10), (200, 50), (512, 512), (513, 10), (700, 10), (1024, 1000), (1025, 1025), (1100, 10),
                                  (2049, 7), (2500, 1500), (4096, 10)])
def test_knn_bitexact_with_gpu_order_oracle(case, oracle, ef, k):
    from cozo_amd.hnsw import HnswSearch
    ids, dist, cnt, nd = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=k, ef=ef), with_n_dist=True)
    oids, odist, ocnt, ond = case["flat"].knn_batch(case["q"], k, ef, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(ids, oids)
    for b in range(len(cnt)):
        assert np.array_equal(dist[b, :cnt[b]], odist[b, :cnt[b]])
    assert int(nd.sum()) == ond, "different number of distance evaluations: traversal differs"


@pytest.mark.parametrize("metric,name", [(1, "Cosine"), (0, "L2")])
def test_small_batches_take_the_speculative_step_with_the_same_results(gpu_lib, oracle, monkeypatch, metric, name):
    """Round 6: a batch that leaves the chip empty -- HnswSearchRA::iter's everyday case, often ONE parent tuple
    (query/ra.rs:1085-1121) -- runs search_level_spec: every live neighbour of a link row is evaluated while the visited test is in
    flight, the next candidate's link row is fetched ahead.  Nothing the reference computes changes: ids, f64 distances, counts and
    the evaluation count n_dist (fresh neighbours only) equal the oracle's and the plain step's, for 1 .. 100 queries, several ef,
    a filter's k = ef, and a radius."""
    from cozo_amd.hnsw import HnswSearch
    dim, m = 768, 16
    x = util.vectors(6000, dim, 5, "lowrank")
    _, flat = util.build_index(oracle, x, metric, m, 48)
    gix = util.gpu_index(flat, name, m)
    qs = util.vectors(100, dim, 6, "lowrank")
    try:
        for B in (1, 3, 64, 100):
            q = qs[:B]
            for ef, k in ((1, 1), (10, 10), (48, 10), (144, 10), (300, 300), (700, 20)):
                oids, odist, ocnt, ond = flat.knn_batch(q, k, ef, dot_mode=oracle.DOT_GPU)
                got = {}
                for spec in ("1", "0"):
                    monkeypatch.setenv("CZ_HNSW_SPEC", spec)
                    ids, dist, cnt, nd = gix.hnsw_knn_batch(q, HnswSearch(k=k, ef=ef), with_n_dist=True)
                    got[spec] = (ids, dist, cnt, nd)
                    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids), (B, ef, spec)
                    for b in range(B):
                        assert np.array_equal(dist[b, :cnt[b]], odist[b, :cnt[b]])
                    assert int(nd.sum()) == ond, (B, ef, spec)
                assert np.array_equal(got["1"][3], got["0"][3])  # per query, too
        monkeypatch.setenv("CZ_HNSW_SPEC", "1")
        r = float(np.median(flat.knn_batch(qs[:8], 10, 64, dot_mode=oracle.DOT_GPU)[1][:, 4]))
        ids, dist, cnt = gix.hnsw_knn_batch(qs[:8], HnswSearch(k=10, ef=64, radius=r))
        oids, odist, ocnt, _ = flat.knn_batch(qs[:8], 10, 64, radius=r, dot_mode=oracle.DOT_GPU)
        assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids)
    finally:
        gix.close()


def test_large_ef_lists_evict_and_shift_over_many_chunks(gpu_lib, oracle):
    """ef up to 4 096 on an index five times that size (VERDICT r3 missing #4: the reference has no limit, hnsw.rs:930-938):
    the LDS list fills, evicts, and the ranked merge shifts it over up to 16 chunks per step; filtered queries keep all ef rows."""
    from cozo_amd.hnsw import HnswSearch
    x = util.vectors(20000, 32, 7, "uniform")
    _, flat = util.build_index(oracle, x, oracle.L2, 12, 60)
    gix = util.gpu_index(flat, "L2", 12)
    q = util.vectors(24, 32, 8, "uniform")
    try:
        for ef, k in [(1500, 10), (2048, 2048), (4096, 100), (4096, 3000)]:
            ids, dist, cnt, nd = gix.hnsw_knn_batch(q, HnswSearch(k=k, ef=ef), with_n_dist=True)
            oids, odist, ocnt, ond = flat.knn_batch(q, k, ef, dot_mode=oracle.DOT_GPU)
            assert np.array_equal(cnt, ocnt) and (cnt == k).all()
            assert np.array_equal(ids, oids), (ef, k)
            assert np.array_equal(dist, odist)
            assert int(nd.sum()) == ond
    finally:
        gix.close()


@pytest.mark.parametrize("metric,name", [(1, "Cosine"), (0, "L2")])
def test_large_ef_pending_buffer_same_results(gpu_lib, oracle, monkeypatch, metric, name):
    """Round 6: from ef = 2 048 on, a 513..768-d search keeps the step's new entries in a sorted pending buffer
    in front of the list and shifts the list only when the buffer is full (hnsw_kernels.h search_level_pending).  Nothing the
    reference computes changes: ids, f64 distances, counts and the evaluation count n_dist equal the oracle's and the plain
    step's (CZ_HNSW_PEND = 0) at ef 2 049 / 4 096 / 8 192 on an index the list does not swallow, with k = ef (a filter's call), a
    radius, and -- Cosine -- an all-zero query whose distances are all NaN (hnsw.rs:575: nothing is `< NaN`).  (The pending form
    exists for two rows in flight per lane group -- a batch that fills the chip; CZ_HNSW_U = 2 selects it for this small batch.)"""
    from cozo_amd.hnsw import HnswSearch
    dim, m = 640, 12
    x = util.vectors(12000, dim, 11, "lowrank")
    _, flat = util.build_index(oracle, x, metric, m, 40)
    gix = util.gpu_index(flat, name, m)
    q = util.vectors(12, dim, 12, "lowrank")
    if metric == 1:
        q[5] = 0.0
    try:
        for ef, k in ((2049, 10), (4096, 4096), (8192, 100)):
            oids, odist, ocnt, ond = flat.knn_batch(q, k, ef, dot_mode=oracle.DOT_GPU)
            per_query = {}
            for pend, u in (("1", "2"), ("0", "2"), ("0", "8")):
                monkeypatch.setenv("CZ_HNSW_PEND", pend)
                monkeypatch.setenv("CZ_HNSW_U", u)
                ids, dist, cnt, nd = gix.hnsw_knn_batch(q, HnswSearch(k=k, ef=ef), with_n_dist=True)
                assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids), (ef, pend, u)
                for b in range(len(cnt)):
                    assert np.array_equal(dist[b, :cnt[b]], odist[b, :cnt[b]], equal_nan=True), (ef, pend, u, b)
                assert int(nd.sum()) == ond, (ef, pend, u)
                per_query[(pend, u)] = nd
            assert np.array_equal(per_query[("1", "2")], per_query[("0", "2")]) and np.array_equal(per_query[("1", "2")], per_query[("0", "8")])
        monkeypatch.setenv("CZ_HNSW_PEND", "1")
        monkeypatch.setenv("CZ_HNSW_U", "2")
        r = float(np.nanmedian(flat.knn_batch(q, 10, 2500, dot_mode=oracle.DOT_GPU)[1][:, 4]))
        ids, dist, cnt = gix.hnsw_knn_batch(q, HnswSearch(k=50, ef=2500, radius=r))
        oids, odist, ocnt, _ = flat.knn_batch(q, 50, 2500, radius=r, dot_mode=oracle.DOT_GPU)
        assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids)
    finally:
        gix.close()


@pytest.mark.parametrize("ef,k", [(64, 10)])
def test_knn_within_tolerance_of_reference_order(case, oracle, ef, k):
    from cozo_amd.hnsw import HnswSearch
    ids, dist, cnt = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=k, ef=ef))
    oids, odist, ocnt, _ = case["flat"].knn_batch(case["q"], k, ef, dot_mode=oracle.DOT_NDARRAY)
    assert np.array_equal(cnt, ocnt)
    same = (ids == oids).all(axis=1)
    assert same.mean() >= 0.95  # near-ties may swap under a different f32 summation order (then the ROWS differ, not the bar)
    # true relative error on every row both orders return (distances below 1e-3 -- a query next to a stored copy of
    # itself -- are pure cancellation: absolute 1e-8 there)
    g, o = dist[same], odist[same]
    big = np.abs(o) >= 1e-3
    if big.any():
        assert np.max(np.abs(g[big] - o[big]) / np.abs(o[big])) <= RTOL
    if (~big).any():
        assert np.max(np.abs(g[~big] - o[~big])) <= 1e-8


@pytest.mark.parametrize("name,metric", [("Cosine", 1), ("IP", 2)])
@pytest.mark.parametrize("n,dim,B", [(1000, 96, 37), (5000, 768, 130), (700, 130, 5)])
def test_bruteforce_gemm_form_matches_sequential_chain_oracle(gpu_lib, oracle, name, metric, n, dim, B):
    """cz_knn_bruteforce(CZ_BF_GEMM): dot products on the matrix cores (v_mfma_f32_32x32x2_f32 = a k-ordered fmaf
    chain per element).  Bit-exact against the oracle in ORC_DOT_SEQ order; ragged tile edges (n, B, dim not
    multiples of the 128 x 128 x 16 tile) included."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    k = 10
    base = util.vectors(n, dim, 11 + dim, "normal")
    q = util.vectors(B, dim, 12 + dim, "normal")
    man = HnswIndexManifest(vec_dim=dim, distance=name, m_neighbours=8)
    ix = GpuHnswIndex(man, base, [None], [np.full((n, 16), 0xFFFFFFFF, dtype=np.uint32)], 0)
    ids, dist = ix.bruteforce_knn(q, k, gemm=True)
    oids, odist = oracle.bruteforce_knn(metric, base, q, k, dot_mode=oracle.DOT_SEQ)
    assert np.array_equal(ids, oids) and np.array_equal(dist, odist)
    # and it is the same neighbourhood the streaming form finds, up to near-ties of the two summation orders
    sids, sdist = ix.bruteforce_knn(q, k)
    assert (ids == sids).all(axis=1).mean() >= 0.95
    assert np.max(np.abs(dist - sdist) / np.maximum(np.abs(sdist), 1e-3)) <= 1e-4 or (ids != sids).any()
    ix.close()
    with pytest.raises(Exception):
        l2 = GpuHnswIndex(HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=8), base, [None],
                          [np.full((n, 16), 0xFFFFFFFF, dtype=np.uint32)], 0)
        l2.bruteforce_knn(q, k, gemm=True)


@pytest.mark.parametrize("name,metric", [("Cosine", 1), ("IP", 2)])
@pytest.mark.parametrize("slab,k", [(256, 10), (1024, 3), (128, 100)])
def test_bruteforce_gemm_fused_selection_stages(gpu_lib, oracle, monkeypatch, name, metric, slab, k):
    """the exhaustive scan with the selection in the GEMM's epilogue (knn_gemm.hip): a first stretch of `slab` columns sets the
    thresholds, later stretches keep only the products that can still enter a list.  A small first stretch (CZ_BF_SLAB) makes a
    small corpus go through several fused stages; sorted-towards-the-query data overflows a stage's candidate buffer and takes
    the fallback.  Bit-exact against the oracle (ORC_DOT_SEQ) either way, and equal to the unfused form (CZ_BF_FUSE = 0)."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    n, dim, B = 9000, 48, 70
    base = util.vectors(n, dim, 51, "normal")
    q = util.vectors(B, dim, 52, "normal")
    base[17] = 0.0  # a zero vector: Cosine NaN, passes every filter and sorts last
    man = HnswIndexManifest(vec_dim=dim, distance=name, m_neighbours=8)
    monkeypatch.setenv("CZ_BF_SLAB", str(slab))
    for data in (base, base[np.argsort(-(base @ q[0]))[::-1]].copy()):  # random order; then ascending score for query 0: every column beats the list
        ix = GpuHnswIndex(man, data, [None], [np.full((n, 16), 0xFFFFFFFF, dtype=np.uint32)], 0)
        ids, dist = ix.bruteforce_knn(q, k, gemm=True)
        oids, odist = oracle.bruteforce_knn(metric, data, q, k, dot_mode=oracle.DOT_SEQ)
        assert np.array_equal(ids, oids) and np.array_equal(dist, odist, equal_nan=True)
        monkeypatch.setenv("CZ_BF_FUSE", "0")
        uids, udist = ix.bruteforce_knn(q, k, gemm=True)
        monkeypatch.delenv("CZ_BF_FUSE")
        assert np.array_equal(ids, uids) and np.array_equal(dist, udist, equal_nan=True)
        ix.close()


def test_search_is_reentrant_on_a_shared_index(case):
    """index handles are immutable after creation and shared by concurrent readers (one SessionTx per thread in the
    reference): four host threads search the same handle at once, every result equals the single-threaded one."""
    import threading
    from cozo_amd.hnsw import HnswSearch
    gix, q = case["gix"], case["q"]
    cfgs = [HnswSearch(k=10, ef=40), HnswSearch(k=5, ef=16), HnswSearch(k=10, ef=120), HnswSearch(k=1, ef=1)]
    serial = [gix.hnsw_knn_batch(q, c, with_n_dist=True) for c in cfgs]
    results, errors = [None] * len(cfgs), []

    def runner(i):
        try:
            for _ in range(4):
                results[i] = gix.hnsw_knn_batch(q, cfgs[i], with_n_dist=True)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=runner, args=(i,)) for i in range(len(cfgs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for got, want in zip(results, serial):
        for a, b in zip(got, want):
            assert np.array_equal(a, b, equal_nan=True)


def test_knn_radius_and_filter_width(case, oracle):
    from cozo_amd.hnsw import HnswSearch
    ef, k = 50, 10
    _, d0, _ = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=k, ef=ef))
    r = float(np.median(d0[:, 4]))
    ids, dist, cnt = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=k, ef=ef, radius=r))
    oids, odist, ocnt, _ = case["flat"].knn_batch(case["q"], k, ef, radius=r, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids)
    assert (cnt < k).any() and (cnt > 0).any()
    for b in range(len(cnt)):
        assert (dist[b, :cnt[b]] <= r).all() and (ids[b, cnt[b]:] == 0xFFFFFFFF).all()
    # a filtered query keeps all ef candidates (hnsw.rs:943-947): k = ef rows come back
    ids2, _, cnt2 = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=k, ef=ef, has_filter=True))
    oids2, _, ocnt2, _ = case["flat"].knn_batch(case["q"], ef, ef, dot_mode=oracle.DOT_GPU)
    assert ids2.shape[1] == ef and np.array_equal(ids2, oids2) and np.array_equal(cnt2, ocnt2)


def test_bruteforce_knn_matches_oracle(case, oracle):
    ids, dist = case["gix"].bruteforce_knn(case["q"], 10)
    oids, odist = oracle.bruteforce_knn(case["metric"], case["x"], case["q"], 10, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids, oids) and np.array_equal(dist, odist)


def test_recall_sanity(case, oracle):
    from cozo_amd.hnsw import HnswSearch
    gt, _ = case["gix"].bruteforce_knn(case["q"], 10)
    ids, _, _ = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=10, ef=200))
    rec = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(gt))])
    assert rec >= 0.8


def test_edge_cases(gpu_lib, oracle):
    from cozo_amd import _lib
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    man = HnswIndexManifest(vec_dim=8, distance="L2", m_neighbours=4)
    # empty index -> no rows (hnsw.rs:903-909)
    e = GpuHnswIndex(man, np.zeros((0, 8), np.float32), [], [], 0xFFFFFFFF)
    ids, dist, cnt = e.hnsw_knn_batch(np.ones((3, 8), np.float32), HnswSearch(k=5, ef=10))
    assert (cnt == 0).all() and (ids == 0xFFFFFFFF).all()
    # single vector, k larger than the index
    x = util.vectors(1, 8, 1)
    _, flat = util.build_index(oracle, x, 0, 4, 10)
    g = util.gpu_index(flat, "L2", 4)
    ids, dist, cnt = g.hnsw_knn_batch(x, HnswSearch(k=5, ef=10))
    assert cnt[0] == 1 and ids[0, 0] == 0 and dist[0, 0] == 0.0 and (ids[0, 1:] == 0xFFFFFFFF).all()
    # dimension mismatch is an error (hnsw.rs:876-878)
    with pytest.raises(ValueError):
        g.hnsw_knn_batch(np.ones((1, 9), np.float32), HnswSearch(k=1, ef=1))
    # ef = 5 000 runs since round 4 (the list is bounded by LDS alone: ~11 000 entries next to a small query) ...
    ids5, _, cnt5 = g.hnsw_knn_batch(x, HnswSearch(k=1, ef=5000))
    assert cnt5[0] == 1 and ids5[0, 0] == 0
    # ... and an ef beyond what 160 KiB of LDS hold is refused loudly, not silently clamped
    with pytest.raises(_lib.CozoGpuError):
        g.hnsw_knn_batch(x, HnswSearch(k=1, ef=20000))
    # duplicates + a zero vector under cosine (NaN distances sort last, never beat a finite distance)
    y = util.vectors(200, 16, 5, "normal")
    y[10] = y[3]
    y[77] = 0.0
    _, flat = util.build_index(oracle, y, 1, 6, 30)
    gc = util.gpu_index(flat, "Cosine", 6)
    ids, dist, cnt = gc.hnsw_knn_batch(y[:32], HnswSearch(k=8, ef=40))
    oids, odist, ocnt, _ = flat.knn_batch(y[:32], 8, 40, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(np.nan_to_num(dist, nan=-1.0), np.nan_to_num(odist, nan=-1.0))
    assert np.array_equal(ids, oids)


@pytest.mark.parametrize("env", [{"CZ_HNSW_VISITED": "hash"}, {"CZ_HNSW_VISITED": "bitmap"},
                                 {"CZ_HNSW_VISITED": "hash", "CZ_HNSW_VSLOTS": "64"},
                                 {"CZ_HNSW_VISITED": "hash", "CZ_HNSW_VSLOTS": "512"}],
                         ids=["hash", "bitmap", "overflow-at-once", "overflow-midway"])
def test_visited_set_forms_are_the_same_set(case, oracle, monkeypatch, env):
    """The per-query visited set is a hash table of node ids that spills into the query's bitmap row when it fills up
    (hnsw_kernels.h VisitedDev).  Every form is an exact set: ids, distances and the evaluation count stay those of the
    oracle -- forced hash tables on these small indices, bitmap only, and tables so small that the move to the bitmap
    happens on the first expansion / in the middle of the level-0 search.  Run twice: the workspace must come back clean."""
    from cozo_amd.hnsw import HnswSearch
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    ef, k = 120, 10
    oids, odist, ocnt, ond = case["flat"].knn_batch(case["q"], k, ef, dot_mode=oracle.DOT_GPU)
    for _ in range(2):
        ids, dist, cnt, nd = case["gix"].hnsw_knn_batch(case["q"], HnswSearch(k=k, ef=ef), with_n_dist=True)
        assert np.array_equal(ids, oids) and np.array_equal(cnt, ocnt) and np.array_equal(dist, odist)
        assert int(nd.sum()) == ond


def test_index_upload_rejects_links_to_missing_nodes(gpu_lib):
    """cz_hnsw_index_create walks every link: an id >= n, or (above level 0) a node that is not on that level, would be
    fetched and inserted into the visited set unchecked by the kernels."""
    from cozo_amd import _lib
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    man = HnswIndexManifest(vec_dim=4, distance="L2", m_neighbours=2)
    x = np.zeros((5, 4), np.float32)
    nb0 = np.full((5, 4), 0xFFFFFFFF, dtype=np.uint32)
    nb0[0, 0] = 1
    GpuHnswIndex(man, x, [None], [nb0], 0).close()
    bad = nb0.copy()
    bad[2, 1] = 5
    with pytest.raises(_lib.CozoGpuError):
        GpuHnswIndex(man, x, [None], [bad], 0)
    # level 1 holds nodes 1 and 3; a level-1 link to node 2 (level 0 only) is refused
    nodes1 = np.array([1, 3], dtype=np.uint32)
    nb1 = np.full((2, 2), 0xFFFFFFFF, dtype=np.uint32)
    nb1[0, 0] = 3
    GpuHnswIndex(man, x, [None, nodes1], [nb0, nb1], 1).close()
    nb1[1, 0] = 2
    with pytest.raises(_lib.CozoGpuError):
        GpuHnswIndex(man, x, [None, nodes1], [nb0, nb1], 1)


def test_filtered_search_predicates_on_the_device(case, oracle):
    """cz_hnsw_search_filtered: the kernel's output stage evaluates `column OP constant` terms on ALL ef candidates and
    returns the first k survivors -- the reference's order of operations (hnsw.rs:943-947 keep ef, :951-1006 radius cut,
    filter, truncate to k).  Checked against the oracle's ef candidates filtered on the host with the reference's
    comparison semantics (Int/Int, Float/Float by total order incl. -0.0 and NaN, mixed as f64)."""
    from cozo_amd.hnsw import HnswSearch, _compare
    gix, flat, q = case["gix"], case["flat"], case["q"]
    n = flat.n
    rng = np.random.default_rng(99)
    col_i = rng.integers(-50, 50, n).astype(np.int64)
    col_f = rng.standard_normal(n)
    col_f[::97] = -0.0
    col_f[5::131] = np.nan
    ci, cf = gix.upload_column(col_i), gix.upload_column(col_f)
    ef, k = 80, 7
    oids, odist, ocnt, ond = flat.knn_batch(q, ef, ef, dot_mode=oracle.DOT_GPU)
    d_med = float(np.median(odist[:, ef // 2]))
    cases = [
        ([(ci, ">=", 10)], None),
        ([(ci, "<", 0), (cf, ">", -0.25)], None),
        ([(cf, "<", 0.0)], None),          # Float vs Float: total order, -0.0 < 0.0 holds
        ([(cf, "==", float("nan"))], None),  # ... and NaN == NaN
        ([(ci, "!=", 3.0), (cf, "<=", 1)], d_med),  # mixed pairs compare as f64; with a radius
        ([(ci, ">", 1000)], None),          # nothing passes
    ]
    cols = {id(ci): col_i, id(cf): col_f}
    for preds, radius in cases:
        ids, dist, cnt, nd = gix.hnsw_knn_batch_filtered(q, HnswSearch(k=k, ef=ef, radius=radius), preds, with_n_dist=True)
        assert int(nd.sum()) == ond
        for b in range(q.shape[0]):
            keep = []
            for j in range(int(ocnt[b])):
                node, d = int(oids[b, j]), float(odist[b, j])
                if radius is not None and d > radius:
                    continue
                vals = [(cols[id(c)][node], op, v) for c, op, v in preds]
                if all(_compare(int(x) if isinstance(x, np.integer) else float(x), op, v) for x, op, v in vals):
                    keep.append((node, d))
                if len(keep) == k:
                    break
            assert cnt[b] == len(keep)
            assert ids[b, :len(keep)].tolist() == [a for a, _ in keep]
            assert dist[b, :len(keep)].tolist() == [d for _, d in keep]
            assert (ids[b, len(keep):] == 0xFFFFFFFF).all()
    ci.close()
    cf.close()


# ---- F64 vectors (VecElementType::F64): VectorCache::dist's F64 arms and hnsw_knn over them --------------
@pytest.mark.parametrize("dim", [1, 2, 7, 8, 9, 31, 33, 64, 128, 130, 768, 1536])
@pytest.mark.parametrize("name,metric", METRICS)
def test_distance_batch_f64_matches_oracle(gpu_lib, oracle, dim, name, metric):
    """bit-equal to the oracle's restatement of the kernel's summation tree (two doubles per chunk), and within 1e-12 of the
    reference's arithmetic (ndarray's unrolled_dot in f64) wherever the value is not the difference of much larger terms"""
    from cozo_amd.hnsw import distance_batch_f64
    rng = np.random.default_rng(dim * 11 + metric)
    base = rng.standard_normal((200, dim))
    q = rng.standard_normal((13, dim))
    base[5] = 0.0  # a zero vector: cosine -> NaN (hnsw.rs:86-95 divides by sqrt(0))
    pairs = np.stack([rng.integers(0, 13, 1500), rng.integers(0, 200, 1500)], 1).astype(np.uint32)
    got = distance_batch_f64(name, base, q, pairs)
    exact = oracle.distance_pairs_f64(metric, base, q, pairs, oracle.DOT_GPU)
    ref = oracle.distance_pairs_f64(metric, base, q, pairs, oracle.DOT_NDARRAY)
    assert np.array_equal(got, exact, equal_nan=True), "kernel summation tree differs from its CPU restatement"
    a, b = q[pairs[:, 0]], base[pairs[:, 1]]
    mag = {0: np.sum((a - b) ** 2, axis=1), 1: np.ones(len(ref)), 2: 1.0 + np.sum(np.abs(a * b), axis=1)}[metric]
    finite = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.max(np.abs(got - ref)[finite] / np.maximum(mag[finite], 1e-300)) <= 1e-12


@pytest.mark.parametrize("dim,dist,metric", [(7, "L2", 0), (128, "Cosine", 1), (768, "Cosine", 1), (33, "IP", 2)])
def test_hnsw_knn_f64_index_matches_oracle(gpu_lib, oracle, dim, dist, metric):
    """an index of f64 vectors searched on the device: ids, f64 distances, counts and evaluation counts equal the oracle's
    hnsw_knn over the same tables with the F64 arms of VectorCache::dist (kernel summation order); f32 queries are converted
    to the index' element type (hnsw.rs:879-884); radius and device-side predicates work as on an f32 index; the f32 entry
    points refuse the handle, and so do build / insert / remove / the exhaustive scan."""
    from cozo_amd import _lib
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, m = 3000, 8
    rng = np.random.default_rng(dim)
    x = rng.standard_normal((n, dim))            # f64 vectors that are NOT representable in f32
    b = oracle.HnswBuilder(dim, metric, m, 40)
    b.insert(x.astype(np.float32), oracle.random_levels(n, m, 3))   # any graph will do: the tables of an f32 build
    f32flat = b.export()
    flat = oracle.FlatIndex(x, metric, f32flat.level_nodes, f32flat.level_nbrs, f32flat.entry, f64=True)
    man = HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m, dtype="F64")
    g = GpuHnswIndex(man, x, [None] + flat.level_nodes[1:], flat.level_nbrs, flat.entry)
    q = rng.standard_normal((40, dim))
    for k, ef in [(10, 32), (1, 1), (5, 200)]:
        ids, dd, cnt, nd = g.hnsw_knn_batch(q, HnswSearch(k=k, ef=ef), with_n_dist=True)
        oids, odd, ocnt, ond = flat.knn_batch(q, k, ef, dot_mode=oracle.DOT_GPU)
        assert np.array_equal(ids, oids) and np.array_equal(dd, odd) and np.array_equal(cnt, ocnt)
        assert int(nd.sum()) == ond
        rids, rdd, _, _ = flat.knn_batch(q, k, ef, dot_mode=oracle.DOT_NDARRAY)
        same = (ids == rids).all(axis=1)
        assert same.mean() >= 0.9
        assert np.max(np.abs(dd[same] - rdd[same]) / np.maximum(np.abs(rdd[same]), 1e-9)) <= 1e-9
    # an f32 query is widened, not the index narrowed
    q32 = q.astype(np.float32)
    ids32, dd32, _ = g.hnsw_knn_batch(q32, HnswSearch(k=5, ef=32))
    oids32, odd32, _, _ = flat.knn_batch(q32.astype(np.float64), 5, 32, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(ids32, oids32) and np.array_equal(dd32, odd32)
    # radius
    r = float(np.median(dd[:, -1]))
    ids_r, dd_r, cnt_r = g.hnsw_knn_batch(q, HnswSearch(k=5, ef=200, radius=r))
    oids_r, odd_r, ocnt_r, _ = flat.knn_batch(q, 5, 200, radius=r, dot_mode=oracle.DOT_GPU)
    assert np.array_equal(cnt_r, ocnt_r) and np.array_equal(ids_r, oids_r)
    # predicates on the device
    col = g.upload_column(np.arange(n, dtype=np.int64))
    fids, fdd, fcnt = g.hnsw_knn_batch_filtered(q, HnswSearch(k=5, ef=64), [(col, "<", n // 2)])
    aids, add_, acnt, _ = flat.knn_batch(q, 64, 64, dot_mode=oracle.DOT_GPU)
    for i in range(len(q)):
        keep = [j for j in range(acnt[i]) if aids[i, j] < n // 2][:5]
        assert fcnt[i] == len(keep) and np.array_equal(fids[i, :len(keep)], aids[i, keep])
    # the other element type's entry points refuse the handle
    L = _lib.lib()
    out_i = np.empty((1, 1), np.uint32); out_d = np.empty((1, 1), np.float64); out_c = np.empty(1, np.uint32)
    rc = L.cz_hnsw_search_batch(g._h, _lib.ptr(q32[:1].copy()), 1, 1, 1, 0, 0.0, _lib.ptr(out_i), _lib.ptr(out_d), _lib.ptr(out_c), None, None, 0, None)
    assert rc == _lib.CZ_E_INVALID
    with pytest.raises(_lib.CozoGpuError):
        g.insert(x[:1].astype(np.float32))
    with pytest.raises(_lib.CozoGpuError):
        g.bruteforce_knn(q32, 3)
    g32 = GpuHnswIndex(HnswIndexManifest(vec_dim=dim, distance=dist, m_neighbours=m), x.astype(np.float32), [None] + flat.level_nodes[1:],
                       flat.level_nbrs, flat.entry)
    rc = L.cz_hnsw_search_batch_f64(g32._h, _lib.ptr(q[:1].copy()), 1, 1, 1, 0, 0.0, _lib.ptr(out_i), _lib.ptr(out_d), _lib.ptr(out_c), None, None, 0, None)
    assert rc == _lib.CZ_E_INVALID
    if dim == 7:
        # the shim's way to the same handle: the index as its stored `tbl:idx` rows + the base relation's f64 vectors, through
        # libcozo_ingest (czi_hnsw_ingest_f64) and cz_hnsw_index_create_f64 -- same rows back as from the arrays above
        from cozo_amd import codec
        from cozo_amd.ingest import StoredHnswIndex, index_relation_tuples
        rows = [(i, x[i]) for i in range(n)]
        tuples = index_relation_tuples([(i, 1, -1) for i in range(n)], x, flat.level_nodes, flat.level_nbrs, flat.entry,
                                       lambda pairs: oracle.distance_pairs_f64(metric, x, x, pairs, oracle.DOT_NDARRAY), relation_id=8)
        stored = StoredHnswIndex(codec.StoredRows.from_tuples(8, tuples, 7), codec.StoredRows.from_tuples(7, rows, 1), [1], dim, metric, m,
                                 dtype="F64")
        gs = stored.to_gpu(man)
        ids_s, dd_s, cnt_s = gs.hnsw_knn_batch(q, HnswSearch(k=10, ef=32))
        ids_a, dd_a, cnt_a = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=32))
        assert np.array_equal(ids_s, ids_a) and np.array_equal(dd_s, dd_a) and np.array_equal(cnt_s, cnt_a)


def test_settle_changes_placement_not_results(gpu_lib, oracle):
    """cz_hnsw_index_settle (placement by trial): the vector table and the visited workspaces are given other places in device memory
    and the faster landing is kept -- ids, distances and counts of a search are the same before and after, whichever candidates
    won; trials = 0 only reports; small indices are not settled by create."""
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    n, dim, m = 6000, 48, 8
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = GpuHnswIndex.build(HnswIndexManifest(vec_dim=dim, distance="L2", m_neighbours=m, ef_construction=40), x, seed=5, max_batch=256)
    q = rng.standard_normal((300, dim)).astype(np.float32)
    assert g.settle() == (0.0, 0.0, 0)  # 1 MB of vectors: create / build leave it alone
    before = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=64), with_n_dist=True)
    import os
    os.environ["CZ_TABLE_SETTLE_TARGET"] = "2.0"  # out of reach: every round is played
    for trials in (1, 3):
        ms0, ms1, tried = g.settle(ef=64, trials=trials)
        assert tried <= 6 * trials and ms0 > 0 and ms1 > 0, (ms0, ms1, tried)
        after = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=64), with_n_dist=True)
        for a, b in zip(before, after):
            assert np.array_equal(a, b)
    ref400 = oracle_free = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=400))  # a larger workspace than the settled one: allocated on demand
    g.settle(ef=400, trials=1)
    again400 = g.hnsw_knn_batch(q, HnswSearch(k=10, ef=400))
    for a, b in zip(ref400, again400):
        assert np.array_equal(a, b)
    os.environ.pop("CZ_TABLE_SETTLE_TARGET")

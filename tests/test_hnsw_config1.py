"""configs[0] of BASELINE.json -- the reference's own CPU-runnable case: N = 10 000, d = 128, f32 ~ U[0,1)
(the reference's rand_vec distribution, data/functions.rs:2149-2156) seed 42, L2, m = 16, ef_construction = 100, 1 000 fresh
queries seed 43, k = 10, ef swept over {16, 32, 64, 128}.  On CPU: the oracle (index build exactly as hnsw_put, search as
hnsw_knn) against the exact scan.  Marked gpu: the same index on the device, bit-exact against the kernel-order oracle for
every query at every ef, and within tolerance / recall of the reference's summation order."""
import numpy as np
import pytest

from tests import util

N, DIM, M, EFC, NQ, K = 10_000, 128, 16, 100, 1_000, 10
EFS = (16, 32, 64, 128)


@pytest.fixture(scope="module")
def c1(oracle):
    x = np.random.default_rng(42).random((N, DIM), dtype=np.float32)
    q = np.random.default_rng(43).random((NQ, DIM), dtype=np.float32)
    builder, flat = util.build_index(oracle, x, oracle.L2, M, EFC)
    truth, _ = oracle.bruteforce_knn(oracle.L2, x, q, K)
    return dict(x=x, q=q, flat=flat, truth=truth, builder=builder)


def recall(ids, truth):
    return float(np.mean([len(set(ids[i].tolist()) & set(truth[i].tolist())) / K for i in range(len(truth))]))


def test_config1_on_the_oracle(c1, oracle):
    flat = c1["flat"]
    # the recall ground truth itself, against float64 numpy: the same neighbour sets up to f32-level near-ties
    d64 = ((c1["q"][:50, None, :].astype(np.float64) - c1["x"][None, :, :].astype(np.float64)) ** 2).sum(-1)
    exact = np.argsort(d64, axis=1, kind="stable")[:, :K]
    assert np.mean([len(set(exact[i].tolist()) & set(c1["truth"][i].tolist())) / K for i in range(50)]) >= 0.998
    assert flat.n == N and flat.level_width[0] == 2 * M and all(w == M for w in flat.level_width[1:])
    assert flat.level_size[0] == N and all(a > b for a, b in zip(flat.level_size, flat.level_size[1:]))
    rec, evals = [], []
    for ef in EFS:
        ids, dist, cnt, nd = flat.knn_batch(c1["q"], K, ef)
        assert np.all(cnt == min(K, ef)) and np.all(np.diff(dist, axis=1) >= 0)
        rec.append(recall(ids, c1["truth"]))
        evals.append(nd / NQ)
    assert all(b > a for a, b in zip(rec, rec[1:])) and all(b > a for a, b in zip(evals, evals[1:]))
    assert rec[-1] >= 0.85  # uniform 128-d data is the hard case: 0.90 at ef = 128 with these parameters
    assert evals[-1] < N / 3  # and the traversal still touches a fraction of the vectors


@pytest.mark.gpu
def test_config1_on_the_device(c1, oracle, gpu_lib):
    from cozo_amd.hnsw import HnswSearch
    gix = util.gpu_index(c1["flat"], "L2", M)
    try:
        for ef in EFS:
            ids, dist, cnt, nd = gix.hnsw_knn_batch(c1["q"], HnswSearch(k=K, ef=ef), with_n_dist=True)
            oids, odist, ocnt, ond = c1["flat"].knn_batch(c1["q"], K, ef, dot_mode=oracle.DOT_GPU)
            assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(dist, odist)
            assert int(nd.sum()) == ond  # the same traversal, distance evaluation for distance evaluation
            rids, rdist, _, _ = c1["flat"].knn_batch(c1["q"], K, ef)  # the reference's (ndarray) summation order
            same = (ids == rids).all(axis=1)
            assert same.mean() >= 0.9  # near-ties may swap under a different f32 summation order
            assert np.max(np.abs(dist[same] - rdist[same]) / np.maximum(np.abs(rdist[same]), 1e-3)) <= 1e-5
            assert abs(recall(ids, c1["truth"]) - recall(rids, c1["truth"])) <= 0.01
    finally:
        gix.close()


# ---- the reference's own tiny index (runtime/tests.rs:743-809 `test_vec_index`) ------------------------------------------------
TINY = [("a", [112, 0]), ("a2", [2, 31]), ("b", [1, 1]), ("b2", [1, 10]), ("bb", [2, 3]), ("bb2", [2, 33]), ("c", [3, 4]), ("c2", [2, 32]),
        ("x", [0, 0.1])]  # the relation after both puts: later rows replaced 'a' and 'b', 'a2'.. were added; key order


def _tiny(oracle):
    x = np.array([v for _, v in TINY], dtype=np.float32)
    builder, flat = util.build_index(oracle, x, oracle.L2, 50, 20)
    return x, flat


def test_reference_tiny_index_known_answer(oracle):
    """`~a:vec{k, v | query: q, k: 2, ef: 20, bind_distance: dist}, q = vec([200, 34])` on the nine 2-d rows of the reference's
    test_vec_index (m: 50, ef_construction: 20, L2).  The reference only prints the rows; with ef = 20 > 9 nodes the level-0
    search reaches the whole index, so the answer is arithmetic: 'a' = [112, 0] at 88^2 + 34^2 = 8900, then 'bb2' = [2, 33] at
    198^2 + 1 = 39205."""
    x, flat = _tiny(oracle)
    ids, dist, cnt, _ = flat.knn_batch(np.array([[200, 34]], dtype=np.float32), 2, 20)
    assert cnt[0] == 2 and [TINY[i][0] for i in ids[0]] == ["a", "bb2"] and dist[0].tolist() == [8900.0, 39205.0]
    assert flat.n_levels >= 1 and flat.level_width[0] == 100


@pytest.mark.gpu
def test_reference_tiny_index_on_the_device(oracle, gpu_lib):
    from cozo_amd.hnsw import HnswSearch
    x, flat = _tiny(oracle)
    gix = util.gpu_index(flat, "L2", 50)
    try:
        ids, dist, cnt = gix.hnsw_knn_batch(np.array([[200, 34]], dtype=np.float32), HnswSearch(k=2, ef=20))
        assert cnt[0] == 2 and [TINY[i][0] for i in ids[0]] == ["a", "bb2"] and dist[0].tolist() == [8900.0, 39205.0]
    finally:
        gix.close()

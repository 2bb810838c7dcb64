"""C++ host mirror on the device: a GPU-built index goes through the store -- GpuHnswIndex::index_rows (the `tbl:idx` rows as
key / value bytes) and GpuHnswIndex::from_stored (libcozo_ingest + cz_hnsw_index_create) -- and answers HnswSearchRA with
the same rows.  (tests/cpp/test_host.cpp, mode gpu-stored.)"""
import pytest

from tests.test_cpp_host import run


@pytest.mark.gpu
def test_cpp_index_through_the_store_gpu(gpu_lib):
    run("gpu-stored")


@pytest.mark.gpu
def test_python_index_through_the_store_gpu(gpu_lib):
    """the same round trip through the Python mirror: GpuHnswIndex.build -> index_rows (stored bytes) -> StoredHnswIndex
    (libcozo_ingest) -> to_gpu: identical tables and identical search results"""
    import numpy as np
    from cozo_amd import build as B, codec
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch
    from cozo_amd.ingest import StoredHnswIndex
    B.build_ingest()
    rng = np.random.default_rng(4)
    n, dim, m = 2000, 32, 8
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=m, ef_construction=40)
    built = GpuHnswIndex.build(man, vecs, seed=3, max_batch=128)
    keys = [(f"k{i:05d}", 1, -1) for i in range(n)]
    idx = built.index_rows(keys, relation_id=12)
    base = codec.StoredRows.from_tuples(11, [(k[0], vecs[i]) for i, k in enumerate(keys)], 1)
    got = StoredHnswIndex(idx, base, [1], dim, 1, m)
    nodes, nbrs, entry = built.export()
    assert got.n == n and got.entry == entry and got.n_levels == len(nbrs)
    assert np.array_equal(got.vectors, vecs)
    for lv in range(len(nbrs)):
        assert np.array_equal(got.level_nodes[lv], nodes[lv]) and np.array_equal(got.level_nbrs[lv], nbrs[lv])
    read = got.to_gpu(man)
    q = rng.standard_normal((64, dim)).astype(np.float32)
    a = built.hnsw_knn_batch(q, HnswSearch(k=10, ef=50))
    b = read.hnsw_knn_batch(q, HnswSearch(k=10, ef=50))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    built.close()
    read.close()

"""C++ host mirror on the device: a GPU-built index goes through the store -- GpuHnswIndex::index_rows (the `tbl:idx` rows as
key / value bytes) and GpuHnswIndex::from_stored (libcozo_ingest + cz_hnsw_index_create) -- and answers HnswSearchRA with
the same rows.  (tests/cpp/test_host.cpp, mode gpu-stored.)"""
import pytest

from tests.test_cpp_host import run


@pytest.mark.gpu
def test_cpp_index_through_the_store_gpu(gpu_lib):
    run("gpu-stored")

"""The multi-GPU entry points of the C ABI on ONE GPU: RCCL communicators of one rank (cz_comm_create_rank with world = 1,
cz_pagerank_multi with n_gpus = 1).  The collectives degenerate, but everything else is the code that runs at N > 1: RCCL
is loaded and initialised, the sharded loop (cozo_amd/csrc/sharded_pagerank.hpp) runs over its HIP backend and its
all-reduces go through RCCL, the sharded search broadcasts / packs / merges.  The exchange logic itself is covered with
world_size 2 in tests/test_sharded_driver.py.  The checks run in a child process (tests/gpu_comm_child.py says why)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_entry_points_on_one_gpu(gpu_lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_comm_child.py")], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    print(p.stdout[-3000:], p.stderr[-3000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "OK pagerank_sharded" in p.stdout and "OK hnsw_search_sharded" in p.stdout and "OK bfs_sharded" in p.stdout
    assert "OK bfs_multi / sssp_multi / connected_components_multi" in p.stdout and "pagerank_sharded_overlapped" in p.stdout
    assert "OK hnsw_multi" in p.stdout
    assert "ALL OK" in p.stdout

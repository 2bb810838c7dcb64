"""The sharded ConnectedComponents driver (cozo_amd/distributed.py) bound to the device entry point, one rank: the
two local passes (own rows, then the star graph of labels) run through cz_connected_components on the GPU.  The
world_size-2 exchange itself is covered on CPU over gloo in tests/test_distributed.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,e", [(5000, 4000), (300, 2000)])
def test_sharded_connected_components_local_passes_on_device(gpu_lib, oracle, n, e):
    from cozo_amd import graph
    from cozo_amd.distributed import sharded_connected_components
    rng = np.random.default_rng(n)
    fi = rng.integers(0, n, e).astype(np.uint32)
    ti = rng.integers(0, n, e).astype(np.uint32)
    off, tgt = oracle.build_csr(n, fi, ti, undirected=True)
    want, want_k = oracle.tarjan_groups(n, off, tgt)
    grp, k = sharded_connected_components(n, fi, ti, 1, torch.device("cpu"), graph.connected_components)
    assert k == want_k and np.array_equal(grp, want)

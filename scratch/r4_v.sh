#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4v
rm -rf $O; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_graph.py -x -q -m gpu -k "ef or wide or edge or probe or search" 2>&1 | tail -5 | tee $O/pytest.txt
echo "== old merge_wide" | tee -a $O/ab.txt
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_old.so timeout 400 python scratch/r4_merge_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/ab.txt
echo "== new merge_wide" | tee -a $O/ab.txt
timeout 400 python scratch/r4_merge_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/ab.txt
python - <<'PY' 2>&1 | tee $O/probe.txt
import sys; sys.path.insert(0, '.')
from cozo_amd import graph as G, _lib
assert _lib.lib().cz_init(0) == 0
for n in (10_000_000, 100_000_000):
    for wb in (4, 8):
        print(n, wb, G.random_access_probe(n, wb))
PY

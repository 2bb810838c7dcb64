#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4bfs
timeout 300 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
echo "== new"; timeout 200 python scratch/r4_bfs.py 2>&1 | grep -v "Warning\|amdgpu" | tee gpurun_out/r4bfs/new.txt
echo "== old"; COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_old.so timeout 200 python scratch/r4_bfs.py 2>&1 | grep -v "Warning\|amdgpu" | tee gpurun_out/r4bfs/old.txt

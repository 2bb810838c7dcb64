#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_sharded_gpu.py tests/test_gpu_hnsw_build.py -q -m gpu 2>&1 | tail -4
bash scratch/r4_o.sh 2>&1 | grep -A40 "bytes" | head -48

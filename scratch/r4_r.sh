#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4r
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python scratch/r4_large_ef.py > $O/large_ef.txt 2>&1; echo "rc=$?"; grep -v "Warning\|amdgpu.ids" $O/large_ef.txt | tail -20
timeout 300 python -m pytest tests/test_gpu_hnsw.py -q -m gpu -k "bitexact or large_ef or edge" 2>&1 | tail -3

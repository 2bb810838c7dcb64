#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python scratch/r4_sssp_repro.py > $O/repro.out 2> $O/repro.err; echo "rc=$?"
grep "^==\|fill dp" $O/repro.err | head -60
timeout 600 python -m pytest tests/test_gpu_hnsw.py -x -q -m gpu -k "bitexact or large_ef" 2>&1 | tail -5

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4d
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python scratch/r4_clock.py > $O/clock.txt 2>&1; echo "clock rc=$?"
grep -v Warning $O/clock.txt | tail -20
timeout 300 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "bfs" 2>&1 | tail -3

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1h
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python scratch/phase_prof.py > $O/phase.txt 2>&1; echo "phase rc=$?"; tail -7 $O/phase.txt
HS_US=0 HS_FLAGS=0,1 HS_REPEAT=2 HS_BS=1024,8192 timeout 600 python scratch/hnsw_sweep.py > $O/hnsw_sweep.txt 2>&1; echo "hnsw_sweep rc=$?"; grep -E "distance_batch|U=|build" $O/hnsw_sweep.txt

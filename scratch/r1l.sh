#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1l
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 600 python scratch/phase_prof.py > $O/phase.txt 2>&1; echo "phase rc=$?"; tail -7 $O/phase.txt
HS_US=0 HS_BS=1024,8192 HS_REPEAT=2 timeout 600 python scratch/hnsw_sweep.py > $O/sweep.txt 2>&1; echo "sweep rc=$?"; grep -E "P=4194304|U=|build" $O/sweep.txt

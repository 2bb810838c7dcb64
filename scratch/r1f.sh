#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1f
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python scratch/phase_prof.py > $O/phase.txt 2>&1; echo "phase rc=$?"; tail -12 $O/phase.txt

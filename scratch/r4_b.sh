#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python scratch/r4_tail.py > $O/tail.txt 2>&1; echo "tail rc=$?"
grep -v Warning $O/tail.txt | tail -20

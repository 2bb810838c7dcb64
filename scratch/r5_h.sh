#!/bin/bash
# round 5, call h: the whole GPU suite, then the automatic choice of the PageRank formulation on both bench graphs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
PR_CFGS=blocked,acc,auto python scratch/r5_pr.py both 2>&1 | grep -v "^/opt" > $O/pr_auto.txt
cut -c1-330 $O/pr_auto.txt

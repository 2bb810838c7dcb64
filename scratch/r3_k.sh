#!/bin/bash
# round 3, call k: plan build stage times, uniform first then R-MAT twice (what is one-time warm-up, what is the skew)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3k; mkdir -p $O
cd $R
CZ_PR_PLAN_TRACE=1 timeout 600 python scratch/r3_pr_rmat.py --only-default --parity 0 --kinds uniform,rmat,rmat,uniform > $O/plan_trace.txt 2>&1; echo "trace rc=$?"; grep -E "\[plan\]|ms/sweep|edges," $O/plan_trace.txt

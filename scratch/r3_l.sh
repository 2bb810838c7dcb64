#!/bin/bash
# round 3, call l: plan build stage times inside bench.py (measure() and the end-to-end first call)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3l; mkdir -p $O
cd $R
CZ_PR_PLAN_TRACE=1 timeout 900 python bench.py --skip-hnsw > $O/bench.json 2> $O/bench.err; echo "rc=$?"; grep -E "\[plan\]|\[bench\] pagerank" $O/bench.err

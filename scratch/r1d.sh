#!/bin/bash
# session c, call 2: random-row fetch ceiling, PageRank XCD-aware block mapping, U=2 default, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1d
rm -rf $O; mkdir -p $O
cd $R
timeout 300 ./scratch/rowfetch_bench 12 > $O/rowfetch.txt 2>&1; echo "rowfetch rc=$?"; cat $O/rowfetch.txt
PR_SWEEP=xcd timeout 300 python scratch/pr_sweep.py > $O/pr_sweep.txt 2>&1; echo "pr_sweep rc=$?"; tail -5 $O/pr_sweep.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; grep -E "built index|ef sweep" $O/bench.err
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

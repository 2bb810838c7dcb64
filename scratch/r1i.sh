#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1i
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
HS_US=0,1 HS_REPEAT=2 HS_BS=1024,8192 timeout 600 python scratch/hnsw_sweep.py > $O/hnsw_sweep.txt 2>&1; echo "hnsw_sweep rc=$?"; grep -E "distance_batch|U=|build" $O/hnsw_sweep.txt
timeout 600 python bench.py --skip-pagerank > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; grep -E "built index|ef sweep" $O/bench.err

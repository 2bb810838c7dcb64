#!/bin/bash
# session c, call 3: DPP butterfly check, gpu tests, hnsw sweep with the pipelined eval, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1e
rm -rf $O; mkdir -p $O
cd $R
timeout 60 ./scratch/dpp_butterfly_test > $O/dpp.txt 2>&1; echo "dpp rc=$?"; cat $O/dpp.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
HS_US=0,1,3 timeout 600 python scratch/hnsw_sweep.py > $O/hnsw_sweep.txt 2>&1; echo "hnsw_sweep rc=$?"; cat $O/hnsw_sweep.txt | tail -16
timeout 600 python bench.py --skip-pagerank > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; grep -E "built index|ef sweep" $O/bench.err

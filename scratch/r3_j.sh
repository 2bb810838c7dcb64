#!/bin/bash
# round 3, call j: comm entry points (overlapped exchange, *_multi traversals), where the R-MAT plan build's time goes, the N > 1 bench code on one rank
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_sharded_gpu.py tests/test_gpu_graph.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
CZ_PR_PLAN_TRACE=1 timeout 600 python scratch/r3_pr_rmat.py --only-default --parity 0 > $O/plan_trace.txt 2>&1; echo "trace rc=$?"; grep -E "\[plan\]|ms/sweep|edges," $O/plan_trace.txt
CZ_BENCH_FORCE_MULTI=1 timeout 600 python bench.py --skip-hnsw --skip-cpu --pr-nodes-total 4000000 --pr-edges-total 40000000 --pr-iters 10 > $O/bench_forced_multi.json 2> $O/bench_forced_multi.err; echo "forced multi rc=$?"; tail -3 $O/bench_forced_multi.err; cut -c1-1500 $O/bench_forced_multi.json

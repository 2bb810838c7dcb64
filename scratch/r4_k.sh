#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4k
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_mirrors_agree.py -q -m gpu 2>&1 | tail -6
timeout 600 python scratch/graph_rules_bench.py > $O/rules.txt 2>&1; echo "rules rc=$?"; tail -25 $O/rules.txt

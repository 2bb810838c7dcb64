#!/bin/bash
# round 2, call k: the device Betweenness rule -- its tests, the tie-rule tests, the C++ host test, and the graph-rules leg of the bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_zz_tie_rules.py tests/test_cpp_host.py tests/test_stored_relation.py tests/test_fixed_rule.py -m gpu -q > gpurun_out/r2k/pytest.txt 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r2k/pytest.txt
ONLY_ALL_SOURCES=1 timeout 600 python scratch/graph_rules_bench.py > gpurun_out/r2k/graph_rules.txt 2>&1
echo "bench rc=$?"; tail -30 gpurun_out/r2k/graph_rules.txt

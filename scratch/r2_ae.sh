#!/bin/bash
# round 2, call ae: SSSP rounds without a memset and with the counters read through pinned memory
O=gpurun_out/r2ae; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_zz_tie_rules.py tests/test_cpp_host.py tests/test_stored_relation.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "rules rc=$?"; grep -E "cz_sssp|cz_closeness|cz_betweenness|all-sources" $O/graph_rules_plain.txt

#!/bin/bash
# round 2, call y: resident graphs (cz_graph_*), the refactored rules, timing of the held forms
O=gpurun_out/r2y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_zz_tie_rules.py tests/test_cpp_host.py tests/test_mirrors_agree.py tests/test_gpu_comm.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "rules rc=$?"; grep -v "amdgpu.ids" $O/graph_rules_plain.txt | tail -18

#!/bin/bash
# round 2, call z: cz_closeness on the device (tests through both mirrors, timing beside the all-sources cz_sssp loop)
O=gpurun_out/r2z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_zz_tie_rules.py tests/test_cpp_host.py tests/test_mirrors_agree.py tests/test_stored_relation.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
ONLY_ALL_SOURCES=1 timeout 600 python scratch/graph_rules_bench.py > $O/all_sources.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/all_sources.txt | tail -5

#!/bin/bash
# round 2, call r: LabelPropagation on the device (tests through both mirrors' paths, time on the 10M / 200M graph)
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py -m gpu -q -k "label or Label" > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -12 $O/pytest.txt
WITH_LP=1 GN=${GN:-10000000} timeout 900 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "rules rc=$?"; grep -v "amdgpu.ids" $O/graph_rules_plain.txt | tail -16

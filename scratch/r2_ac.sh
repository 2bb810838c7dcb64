#!/bin/bash
# round 2, call ac: LabelPropagation with the class lists built on the device and the colouring over a shrinking list
O=gpurun_out/r2ac; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_mirrors_agree.py tests/test_cpp_host.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
WITH_LP=1 timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "rules rc=$?"; grep -E "label_propagation|colour classes" $O/graph_rules_plain.txt

#!/bin/bash
# round 2, call p: SSSP with canonical parents inside the relaxation + device weight check; all graph-side tests; timings
O=gpurun_out/r2p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_stored_relation.py tests/test_cpp_host.py tests/test_zz_tie_rules.py tests/test_gpu_comm.py tests/test_zz_stored_index_cpp_gpu.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "bench rc=$?"; grep -v "amdgpu.ids" $O/graph_rules_plain.txt | tail -14

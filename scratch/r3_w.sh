#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_zz_tie_rules.py tests/test_gpu_comm.py -q -x -k "sssp or dijkstra or closeness or betweenness or tie or shortest" > gpurun_out/r3w/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3w/pytest.txt
timeout 600 python scratch/r3_rule_runs.py sssp 3 2>&1 | grep -v amdgpu.ids | grep "device ms"

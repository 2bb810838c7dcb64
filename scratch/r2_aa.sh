#!/bin/bash
# round 2, call aa: vertex-partitioned ConnectedComponents behind the C ABI (one rank on the device), comm tests
O=gpurun_out/r2aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_graph.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 300 python tests/gpu_comm_child.py > $O/child.txt 2>&1; echo "child rc=$?"; grep -E "^OK|ALL OK|Error|error" $O/child.txt | tail -8

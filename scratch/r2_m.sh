#!/bin/bash
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_graph.py tests/test_cpp_host.py tests/test_zz_tie_rules.py tests/test_zz_stored_index_cpp_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.txt

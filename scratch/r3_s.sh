#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3s
timeout 1500 python -m pytest tests/test_zz_stored_index_cpp_gpu.py tests/test_zz_tie_rules.py tests/test_cpp_host.py -m gpu -q > gpurun_out/r3s/pytest.txt 2>&1
echo "pytest rc=$?"
tail -6 gpurun_out/r3s/pytest.txt

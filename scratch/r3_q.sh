#!/bin/bash
# extend_candidates in the device build: parity tests + the existing build tests
export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
timeout 1500 python -m pytest tests/test_gpu_hnsw_build.py -x -q > gpurun_out/r3q/pytest.txt 2>&1
echo "pytest rc=$?"
tail -25 gpurun_out/r3q/pytest.txt

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3x
timeout 900 python -m pytest tests/test_gpu_comm.py -q -x > gpurun_out/r3x/pytest.txt 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r3x/pytest.txt | cut -c1-300

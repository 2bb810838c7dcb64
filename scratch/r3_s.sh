#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3s
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3s/pytest.txt 2>&1
echo "pytest rc=$?"
tail -6 gpurun_out/r3s/pytest.txt

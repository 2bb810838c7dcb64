#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3final/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3final/pytest_gpu.txt
timeout 600 python -c "
import __graft_entry__ as g
g.smoke()
print('smoke ok')
" 2>&1 | grep -v amdgpu.ids | tail -3

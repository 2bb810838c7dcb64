#!/bin/bash
export TMPDIR=/tmp
for args in "1000 8 1" "1000 8 0" "3000 64 0" "3000 64 1"; do
  timeout 300 python scratch/r3_ext_batch.py $args 2>&1 | grep -v amdgpu.ids | tail -2
done
echo "eager:"
CZ_BUILD_LAZY=0 timeout 300 python scratch/r3_ext_batch.py 3000 64 1 2>&1 | grep -v amdgpu.ids | tail -2

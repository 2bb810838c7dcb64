#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
timeout 600 python scratch/r3_ext_debug.py 2>&1 | grep -v amdgpu.ids | tail -30

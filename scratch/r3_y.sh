#!/bin/bash
export TMPDIR=/tmp
timeout 900 python scratch/r3_sssp_repeat.py 2>&1 | grep -v amdgpu.ids | tail -14

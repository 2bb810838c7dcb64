#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3r
timeout 1500 python scratch/r3_index_quality.py 1000000 768 32 200 clustered 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3r/quality_clustered_1m.txt | cut -c1-330

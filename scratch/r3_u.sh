#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3u
SWEEP_DELTA=1 timeout 600 python scratch/r3_rule_runs.py sssp 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3u/sssp_delta.txt | tail -12
CZ_SSSP_TRACE=1 timeout 600 python scratch/r3_rule_runs.py sssp 1 2>&1 | grep "sssp phase" > gpurun_out/r3u/sssp_trace.txt
wc -l gpurun_out/r3u/sssp_trace.txt; awk '{n+=$9} END {print "entries relaxed in total:", n}' gpurun_out/r3u/sssp_trace.txt; head -30 gpurun_out/r3u/sssp_trace.txt

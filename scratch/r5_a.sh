#!/bin/bash
# round 5, call a: the contiguous vector table (CZ_TABLE_CONTIGUOUS default on) against plain hipMalloc, search kernel, 4M x 768
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
echo "== contiguous (default) ==" > gpurun_out/r5a/probe.txt
HS_N=4000000 timeout 400 python scratch/r5_handle_probe.py >> gpurun_out/r5a/probe.txt 2>&1
echo "== CZ_TABLE_CONTIGUOUS=0 ==" >> gpurun_out/r5a/probe.txt
CZ_TABLE_CONTIGUOUS=0 HS_N=4000000 timeout 400 python scratch/r5_handle_probe.py >> gpurun_out/r5a/probe.txt 2>&1
cat gpurun_out/r5a/probe.txt

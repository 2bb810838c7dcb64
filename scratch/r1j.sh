#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1j
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python scratch/placement.py > $O/placement.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/placement.txt | tail -14

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4u
rm -rf $O; mkdir -p $O
cd $R
PH_DIST=clustered PH_EF=${PH_EF:-768,2048,4096,8192} timeout 600 python scratch/phase_prof.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/phases.txt | tail -60

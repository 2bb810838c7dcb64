#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4w
rm -rf $O; mkdir -p $O
cd $R
for round in 1 2; do
echo "== new (round $round)" | tee -a $O/ab.txt
HS_DIST=lowrank HS_EF=96,144,256 timeout 300 python scratch/r4_merge_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/ab.txt
echo "== old (round $round)" | tee -a $O/ab.txt
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_old.so HS_DIST=lowrank HS_EF=96,144,256 timeout 300 python scratch/r4_merge_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/ab.txt
done

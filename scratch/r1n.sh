#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1n
rm -rf $O; mkdir -p $O
cd $R
rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -6
for v in old default flat_nt0 old default; do
  if [ $v = default ]; then unset COZO_GPU_LIB; else export COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_$v.so; fi
  HS_US=0,1 HS_BS=1024 timeout 600 python scratch/hnsw_sweep.py > $O/sweep_$v.txt 2>&1; echo "== $v rc=$?"; grep -E "U=|build" $O/sweep_$v.txt
done
rocm-smi --showpower --showclocks 2>&1 | grep -iE "sclk|mclk|power" | head -6

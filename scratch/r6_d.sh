#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" w4f6 w8f5; do
  echo "== variant ${v:-default(w8f4)}"
  if [ -n "$v" ]; then export COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_$v.so; else unset COZO_GPU_LIB; fi
  IP_PARITY=${IP_PARITY:-0} IP_CFGS=${IP_CFGS:-t16s16,t16s32,t32s32,t16s16_p8} timeout 900 python scratch/r6_inplace.py uniform 2>&1 | grep -v Warning | grep sweep | cut -c1-170
done

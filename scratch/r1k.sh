#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1k
rm -rf $O; mkdir -p $O
cd $R
timeout 300 ./scratch/rowfetch_bench 4 compute > $O/rowfetch_compute.txt 2>&1; echo "rowfetch rc=$?"; cat $O/rowfetch_compute.txt
for v in default snt1 default snt1; do
  if [ $v = default ]; then unset COZO_GPU_LIB; else export COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_$v.so; fi
  HS_US=0 HS_BS=1024,8192 timeout 600 python scratch/hnsw_sweep.py > $O/sweep_$v.txt 2>&1; echo "== $v rc=$?"; grep -E "P=4194304|U=" $O/sweep_$v.txt
done

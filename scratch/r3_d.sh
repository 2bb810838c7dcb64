#!/bin/bash
# round 3, call d: phase B knobs A/B on one box (runs per wave in flight 10 / 16 / 20, row data requested late, short rows batched), phase cycles of the default
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "long_rows or skewed or blocked_sweep or pagerank_bitexact or sharded_plan" > $O/pytest_pagerank.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_pagerank.txt
for v in default prbase pru20 prrows pru16 default; do
  if [ $v = default ]; then unset COZO_GPU_LIB; else export COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_$v.so; fi
  echo "== $v" >> $O/ab.txt
  timeout 300 python scratch/r3_pr_rmat.py --only-default --parity 0 2>&1 | grep -E "ms/sweep|phase" >> $O/ab.txt
done
unset COZO_GPU_LIB
cat $O/ab.txt
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_prphase.so timeout 600 python scratch/r3_pr_rmat.py --only-default --parity 0 > $O/pr_phase.txt 2>&1; echo "phase rc=$?"; grep -E "phase|ms/sweep" $O/pr_phase.txt
timeout 600 python scratch/r3_pr_rmat.py --kinds rmat > $O/pr_rmat.txt 2>&1; echo "rmat sweep rc=$?"; grep -E "ms/sweep|parity" $O/pr_rmat.txt

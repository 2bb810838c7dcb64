#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1g
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python scratch/phase_prof.py > $O/phase.txt 2>&1; echo "phase rc=$?"; tail -7 $O/phase.txt
HS_US=0,1 timeout 600 python scratch/hnsw_sweep.py > $O/hnsw_sweep.txt 2>&1; echo "hnsw_sweep rc=$?"; grep -E "distance_batch|U=" $O/hnsw_sweep.txt
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_nt0.so HS_US=0 timeout 600 python scratch/hnsw_sweep.py > $O/hnsw_sweep_nt0.txt 2>&1; echo "hnsw_sweep nt0 (old merge) rc=$?"; grep -E "distance_batch|U=" $O/hnsw_sweep_nt0.txt
timeout 600 python bench.py --skip-pagerank > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; grep -E "built index|ef sweep" $O/bench.err

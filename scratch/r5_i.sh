#!/bin/bash
# round 5, call i: batch-size ladder of hnsw_knn_kernel, product build against the round-4 library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
python scratch/r5_ladder.py 2>&1 | grep -v "^/opt" > $O/ladder_product.txt
cat $O/ladder_product.txt
HS_CHECK=0 COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_r4.so python scratch/r5_ladder.py 2>&1 | grep -v "^/opt" > $O/ladder_r4.txt
cat $O/ladder_r4.txt

#!/bin/bash
# round 6: (1) the Jacobi reading in the grouped-tile layout (CZ_PR_INPLACE_AS_JACOBI) on the uniform and the R-MAT graph, beside the
# product formulations; (2) VERDICT r5 item 4: value chunks that stay in the Infinity Cache (CZ_PR_CHUNKS, blocked formulation)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6g
rm -rf $O; mkdir -p $O
cd $R
IP_CFGS=jacobi_t16s16,jacobi_t16s32,jacobi_t32s32,jacobi_t16s32_p64 timeout 900 python scratch/r6_inplace.py uniform 2>&1 | grep -v Warning | cut -c1-400 > $O/jacobi_grouped_uniform.txt; tail -5 $O/jacobi_grouped_uniform.txt | cut -c1-200
IP_CFGS=jacobi_t16s16,jacobi_t16s32,jacobi_t32s32,t16s16 timeout 900 python scratch/r6_inplace.py rmat 2>&1 | grep -v Warning | cut -c1-400 > $O/jacobi_grouped_rmat.txt; tail -5 $O/jacobi_grouped_rmat.txt | cut -c1-200
PR_CFGS=blocked,blocked_c2,blocked_c3,blocked_c4,blocked_c8,acc timeout 900 python scratch/r5_pr.py uniform 2>&1 | grep -v Warning | cut -c1-220 > $O/pagerank_mall.txt; cat $O/pagerank_mall.txt | cut -c1-160

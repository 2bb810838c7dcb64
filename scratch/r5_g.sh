#!/bin/bash
# round 5, call g: phases of pa_reduce (profiling build), then tests + kernel trace + both graphs with the product build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_prphase.so PR_CFGS=acc python scratch/r5_pr.py uniform 2>&1 | grep -v "^/opt" | cut -c1-150
bash scratch/r5_e.sh 2>&1 | grep -v "^W2026\|^E2026" | grep "passed\|failed\|pa_reduce\|pb_expand\|blocked \|acc" | cut -c1-150

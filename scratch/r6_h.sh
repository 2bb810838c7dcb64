#!/bin/bash
# round 6: kernel trace of the five rules on the R-MAT graph (which kernels the skewed graph's time sits in)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6h
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o rules -- python $R/scratch/r6_rules.py rmat > $O/run.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/rules_rmat_kernel_stats.txt; head -45 $O/rules_rmat_kernel_stats.txt | cut -c1-150
rm -rf $O/trace

#!/bin/bash
# round 6, part G: kernel trace of the R-MAT rule leg alone (which BFS kernels hold the 11.9 ms)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6g
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/scratch/r6_rules.py rmat > $O/rules.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/rmat_kernel_stats.txt
grep -E "^bfs|^scan|^sssp" $O/rmat_kernel_stats.txt | cut -c1-150
grep -E "^bfs |^sssp " $O/rules.txt | cut -c1-120
rm -rf $O/trace

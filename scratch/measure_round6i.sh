#!/bin/bash
# round 6, part I: per-level durations of the BFS kernels on the R-MAT graph
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6i
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/scratch/r6_rules.py rmat > $O/rules_rmat.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/scratch/bfs_levels_trace.py "$db" > $O/bfs_levels.txt 2>&1
cat $O/bfs_levels.txt | tail -90
rm -rf $O/trace

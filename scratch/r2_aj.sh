#!/bin/bash
# round 2, call aj: kernel trace of the graph rules on the final tree (10M / 100M graph + the 20k-node all-sources rules)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r2aj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WITH_LP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o gr -- python $R/scratch/graph_rules_bench.py > $R/$O/graph_rules.txt 2>&1
echo "rc=$?"
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; head -48 $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

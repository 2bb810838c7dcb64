#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/graph_rules
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o g -- python $R/scratch/graph_rules_bench.py > $O/out.txt 2>&1; echo "rc=$?"
grep -v amdgpu.ids $O/out.txt | tail -12
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/kernel_stats.txt; grep -E "^kernel|bfs_|cc_|sssp|triangles|scan|frontier" $O/kernel_stats.txt | cut -c1-150 | head -24
rm -rf $O/trace

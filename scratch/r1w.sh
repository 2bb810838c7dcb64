#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1w
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --skip-pagerank --skip-cpu --steps 3 > $O/bench.json 2> $O/bench.err
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/kernel_stats.txt; grep -E "gemm|select|norms|bf_|^kernel" $O/kernel_stats.txt | cut -c1-170
grep -E "ground truth" $O/bench.err
rm -rf $O/trace

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1v
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q -k "bruteforce" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --skip-pagerank --skip-cpu --steps 3 > $O/bench.json 2> $O/bench.err
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/kernel_stats.txt; grep -E "gemm|select|norms|bf_|^kernel" $O/kernel_stats.txt | cut -c1-170
grep -E "ground truth|ef sweep" $O/bench.err
rm -rf $O/trace

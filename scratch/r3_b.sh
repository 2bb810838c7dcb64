#!/bin/bash
# round 3, call b: PageRank with length-ordered heavy rows + the leaner exact sum (settings sweep, parity at full size), distance batch in region order
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "long_rows or skewed or blocked_sweep" > $O/pytest_pagerank.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pagerank.txt
timeout 900 python scratch/r3_pr_rmat.py > $O/pr_rmat.txt 2>&1; echo "pr sweep rc=$?"; grep -v Warning $O/pr_rmat.txt | tail -30
timeout 600 python scratch/r3_dist_region.py > $O/dist_region.txt 2>&1; echo "dist rc=$?"; grep -v Warning $O/dist_region.txt | tail -14
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o pr -- python $R/scratch/r3_pr_rmat.py --kinds rmat --parity 0 --only-default > $R/$O/pr_rmat_traced.txt 2>&1
echo "trace rc=$?"
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; grep -E "pb_|pr_|region_|distance_" $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

#!/bin/bash
# round 3, call e: row blocks of a skewed plan dealt to the XCDs by cost instead of by count (R-MAT), knobs back at their measured best (uniform)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "long_rows or skewed or blocked_sweep or pagerank_bitexact or sharded_plan" > $O/pytest_pagerank.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_pagerank.txt
timeout 600 python scratch/r3_pr_rmat.py > $O/pr_rmat.txt 2>&1; echo "sweep rc=$?"; grep -E "ms/sweep|parity" $O/pr_rmat.txt
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_prphase.so timeout 600 python scratch/r3_pr_rmat.py --only-default --parity 0 > $O/pr_phase.txt 2>&1; echo "phase rc=$?"; grep -E "phase|ms/sweep" $O/pr_phase.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o pr -- python $R/scratch/r3_pr_rmat.py --kinds rmat --parity 0 --only-default > $R/$O/pr_rmat_traced.txt 2>&1
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; grep -E "pb_|pr_" $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

#!/bin/bash
# round 2, call b: the hnsw gpu tests on the new traversal, then the A/B of its knobs at 1M, then a kernel trace of the 1M build
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_hnsw_build.py tests/test_hnsw_config1.py tests/test_hnsw_search_ra.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 900 python scratch/r2_hnsw_ab.py > $O/ab.txt 2>&1; echo "ab rc=$?"; cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
HS_LAZY=1 HS_REPEAT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o ab -- python $GRAFT_REPO_ROOT/scratch/r2_hnsw_ab.py > $GRAFT_REPO_ROOT/$O/ab_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/ab_kernel_stats.txt; head -16 $O/ab_kernel_stats.txt | cut -c1-170
rm -rf $O/trace

#!/bin/bash
# round 2, call h: insert/remove + comm + graph tests (exit code!), graph rules under a kernel trace
O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_hnsw_build.py tests/test_gpu_comm.py tests/test_gpu_graph.py tests/test_gpu_hnsw.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o gr -- python $GRAFT_REPO_ROOT/scratch/graph_rules_bench.py > $GRAFT_REPO_ROOT/$O/graph_rules.txt 2>&1
cd $GRAFT_REPO_ROOT
cat $O/graph_rules.txt | grep -v amdgpu.ids
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/graph_rules_kernel_stats.txt; head -24 $O/graph_rules_kernel_stats.txt | cut -c1-170
rm -rf $O/trace

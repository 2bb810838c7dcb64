#!/bin/bash
# round 2, call o: BFS with visited bits + won bytes, level-ordered Betweenness, rule timings by the library's clock; kernel trace
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_stored_relation.py tests/test_cpp_host.py tests/test_zz_tie_rules.py tests/test_gpu_comm.py tests/test_zz_stored_index_cpp_gpu.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "bench rc=$?"; grep -v "amdgpu.ids" $O/graph_rules_plain.txt | tail -14
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o gr -- python $GRAFT_REPO_ROOT/scratch/graph_rules_bench.py > $GRAFT_REPO_ROOT/$O/graph_rules.txt 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; head -30 $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

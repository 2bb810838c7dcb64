#!/bin/bash
# round 2, call l: union-find CC (tests + timing) and where the single-source SSSP's time goes (kernel trace + pile sizes per round)
O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_stored_relation.py tests/test_cpp_host.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
AN=2000 AE=20000 CZ_SSSP_TRACE=1 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o gr -- python $GRAFT_REPO_ROOT/scratch/graph_rules_bench.py > $GRAFT_REPO_ROOT/$O/graph_rules.txt 2> $GRAFT_REPO_ROOT/$O/stderr.txt
echo "bench rc=$?"
cd $GRAFT_REPO_ROOT
grep -v "amdgpu.ids" $O/graph_rules.txt | tail -20
grep "^sssp phase" $O/stderr.txt | head -300 > $O/sssp_rounds.txt; wc -l $O/sssp_rounds.txt; rm -f $O/stderr.txt
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; head -24 $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

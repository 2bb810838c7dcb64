#!/bin/bash
# round 2, call n: SSSP with workgroup-staged pile pushes -- parity tests, the replayed round, wall times, kernel trace
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_stored_relation.py tests/test_cpp_host.py tests/test_zz_tie_rules.py tests/test_gpu_comm.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.txt
CZ_SSSP_EXPERIMENT=1 timeout 600 python scratch/sssp_experiment.py > $O/experiment.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/experiment.txt | tail -24
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o gr -- python $GRAFT_REPO_ROOT/scratch/graph_rules_bench.py > $GRAFT_REPO_ROOT/$O/graph_rules.txt 2>&1
echo "bench rc=$?"
cd $GRAFT_REPO_ROOT
grep -v "amdgpu.ids" $O/graph_rules.txt | tail -20
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; head -24 $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

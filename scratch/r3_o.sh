#!/bin/bash
# round 3, call o: oriented triangle count (tests + time on the 10M / 200M graph), malformed-shard refusal, the single-source SSSP alone under the kernel trace
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3o; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_comm.py tests/test_gpu_hnsw_build.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules.txt 2>&1; echo "rules rc=$?"; grep -v Warning $O/graph_rules.txt | tail -16
CZ_TRI_GENERAL=1 timeout 600 python scratch/graph_rules_bench.py 2>&1 | grep -E "clustering" > $O/tri_general.txt; cat $O/tri_general.txt
CZ_SSSP_TRACE=1 timeout 300 python scratch/r3_rule_runs.py sssp 1 > $O/sssp_rounds.txt 2>&1; grep -c "sssp phase" $O/sssp_rounds.txt; grep "sssp phase" $O/sssp_rounds.txt | awk '{print $3}' | sort -n | uniq -c | sort -k2 -n | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o sssp -- python $R/scratch/r3_rule_runs.py sssp 2 > $R/$O/sssp_traced.txt 2>&1
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/sssp_kernel_stats.txt; grep -E "sssp_|fill_|copyBuffer|fillBuffer" $O/sssp_kernel_stats.txt | cut -c1-170
rm -rf $O/trace

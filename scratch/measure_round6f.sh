#!/bin/bash
# round 6, final tree, part F: after the SSSP long-list queue (graph.hip SsspLongQ) -- the PMC passes of the graph.hip entries, the GPU
# suite, and the R-MAT rule leg with the queue on and off on the same box (profiles/r06_sssp_long_queue.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6f
rm -rf $O; mkdir -p $O
cp $R/profiles/r06_pmc_bench_detail.json $O/bench_detail.json
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag, kernel regex, command...
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$? ($(date +%T))"
  done
}
pmc bfs "bfs_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py bfs 2
pmc sssp "sssp_|fill_u64_kernel" python $R/scratch/r3_rule_runs.py sssp 2
pmc cc "cc_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py cc 2
pmc tri "triangles_|tri_" python $R/scratch/r3_rule_runs.py tri 2
pmc lp "lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py lp 2
grep -h "Traceback\|Error" $O/pmc_*.out | grep -v Warning | head -8
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt | tail -6
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.txt
{
  echo "# R-MAT 10M / 200M rule leg (scratch/r6_rules.py rmat: bench.py's bench_graph_rules), same box, long-list queue on / off / on"
  for q in 1 0 1; do
    echo "## CZ_SSSP_LONG_QUEUE=$q"
    CZ_SSSP_LONG_QUEUE=$q timeout 600 python scratch/r6_rules.py rmat 2>&1 | grep -v Warning | grep -E "^sssp|^bfs|^wall|^\{" | cut -c1-260
  done
  echo "## uniform 10M / 100M (no list beyond 256 edges: no queue is allocated, no second launch)"
  timeout 600 python scratch/r6_rules.py uniform 2>&1 | grep -v Warning | grep -E "^sssp" | cut -c1-260
} > $O/sssp_long_queue.txt 2>&1
cat $O/sssp_long_queue.txt

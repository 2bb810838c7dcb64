#!/bin/bash
# round 6, part D: the in-place PageRank plan alone (after a change to inplace_plan.hpp / pagerank_inplace.hip): its GPU tests, its PMC
# passes, the host build's stage times.  Needs profiles/r06_pmc_bench_detail.json's algorithmic bytes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6d
rm -rf $O; mkdir -p $O
cp $R/profiles/r06_pmc_bench_detail.json $O/bench_detail.json
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -q -k "inplace or in_place" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag, kernel regex, command...
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$? ($(date +%T))"
  done
}
IP_CFGS=t16s16 IP_PARITY=0 IP_FEW=1 pmc prip "gi_level_kernel|gi_long_kernel|gi_sum_partials_kernel" python $R/scratch/r6_inplace.py uniform
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; grep -i "inplace" $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
for i in 1 2 3; do CZ_PLAN_TRACE=1 IP_CFGS=t16s16 IP_PARITY=1 IP_FEW=1 timeout 600 python scratch/r6_inplace.py uniform 2>&1 | grep -v Warning | grep -E "inplace plan|create|parity|ms" | tail -14; done > $O/plan_build.txt; cat $O/plan_build.txt

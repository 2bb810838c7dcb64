#!/bin/bash
# PMC traffic of the remaining whole-graph rules (ConnectedComponents, ClusteringCoefficients, LabelPropagation) on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3pmc
rm -rf $O; mkdir -p $O
cp $R/profiles/r03_bench_detail.json $O/bench_detail.json
cd /tmp; export TMPDIR=/tmp
pmc() {
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$?"
  done
}
finish() {
  cd $R
  python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; grep -v "left as it was" $O/pmc_summary.txt
  cp profiles/pmc_traffic.json $O/pmc_traffic.json
  cd /tmp
}
pmc cc "cc_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py cc 2
pmc tri "triangles_|tri_" python $R/scratch/r3_rule_runs.py tri 2
finish
pmc lp "lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py lp 2
finish
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null

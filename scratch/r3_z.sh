#!/bin/bash
# after the pooled SSSP scratch: tests, the PMC passes of the two graph.hip entries again (source hash), the bench line again
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3z
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_zz_tie_rules.py tests/test_cpp_host.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
cd /tmp
pmc() {
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$?"
  done
}
pmc bfs "bfs_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py bfs 2
pmc sssp "sssp_|fill_u64_kernel" python $R/scratch/r3_rule_runs.py sssp 2
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 1500 python bench.py --skip-cpu > $O/bench_with_traffic.json 2> $O/bench.err; echo "bench rc=$?"
cp gpurun_out/bench_detail.json $O/bench_detail.json
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = json.load(open(R + "/gpurun_out/r3z/bench_detail.json"))
print("hnsw", d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
g = d["graph_rules"]
for k in ("bfs", "connected_components", "sssp"):
    print(k, g[k]["wall_ms"], g[k]["device_ms"], g[k].get("repeated_call_wall_ms"), g[k]["roofline"].get("traffic"))
PY

#!/bin/bash
# round 6, final tree, part J (after the BFS hub-level changes): PMC passes of the graph.hip entries, the GPU suite, smoke, then the
# graph-rule legs of the bench line as `bench.py --skip-hnsw` runs them (PageRank + rule legs with their CPU baselines and parity)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6j
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
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; tail -6 $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-hnsw > $O/bench_graph_legs.json 2> $O/bench_graph_legs.err; echo "bench --skip-hnsw rc=$?"
cp gpurun_out/bench_detail.json $O/bench_detail_graph_legs.json
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = json.load(open(R + "/gpurun_out/round6j/bench_detail_graph_legs.json"))
print("wall", d.get("bench_wall_s"))
for leg in ("graph_rules", "graph_rules_rmat"):
    for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
        o = d.get(leg, {}).get(k, {}); print(leg, k, o.get("device_ms"), o.get("roofline", {}).get("traffic"), o.get("parity_checked"), json.dumps(o.get("cpu_baseline"))[:120])
PY

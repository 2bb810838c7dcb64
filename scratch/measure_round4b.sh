#!/bin/bash
# round 4, second half of the measurement set: after the plan build got its own sort / scan (pagerank.hip) and graph.hip lost the
# experiment code, the PMC traffic of the PageRank sweeps and the graph rules is taken again on the final tree (the HNSW / distance
# entries of profiles/pmc_traffic.json keep their source hashes), then the GPU suite and the full bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round4b
rm -rf $O; mkdir -p $O
cp $R/profiles/r04_bench_detail.json $O/bench_detail.json
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag, kernel regex, command...
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$?"
  done
}
pmc pr "pb_expand_kernel|pb_reduce_kernel|pr_step_kernel" python $R/bench.py --skip-hnsw --skip-cpu --skip-secondary --pr-iters 3
pmc prrmat "pb_expand_kernel|pb_reduce_kernel|pr_hub_kernel|pr_empty_rows_kernel" python $R/scratch/r3_pr_rmat.py --kinds rmat --only-default --parity 0
pmc bfs "bfs_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py bfs 2
pmc sssp "sssp_|fill_u64_kernel" python $R/scratch/r3_rule_runs.py sssp 2
pmc cc "cc_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py cc 2
pmc tri "triangles_|tri_" python $R/scratch/r3_rule_runs.py tri 2
pmc lp "lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py lp 2
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -6
cp gpurun_out/bench_detail.json $O/bench_final_detail.json
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = json.load(open(R + "/gpurun_out/round4b/bench.json"))
print("line bytes", len(open(R + "/gpurun_out/round4b/bench.json").read()), "wall", d["bench_wall_s"])
print("hnsw", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "ceiling", d["roofline"].get("measured_ceiling"))
for k in ("distance_batch", "pagerank", "pagerank_rmat", "hnsw_1m", "hnsw_1m_clustered"):
    o = d.get(k, {}); print("  ", k, o.get("roofline", {}).get("frac"), o.get("roofline", {}).get("traffic"), o.get("ms_per_iteration"), o.get("inplace_reading"))
for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
    o = d.get("graph_rules", {}).get(k, {}); print("  ", k, o.get("wall_ms"), o.get("device_ms"), o.get("roofline", {}).get("traffic"), o.get("repeated_call_wall_ms"), o.get("repeated_call_laps_ms"))
print("   box", json.dumps(d.get("box"))[:700])
PY

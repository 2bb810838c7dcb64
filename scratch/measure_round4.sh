#!/bin/bash
# round 4's measurement set: the GPU suite, the plain bench line (box telemetry + measured HBM ceilings inside), the bench under rocprofv3
# kernel trace, the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs, no tracing) for every workload whose `roofline.traffic`
# the line reports, and the line again with the traffic filled in.  Copy gpurun_out/round4/* to profiles/r04_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round4
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -12
cp gpurun_out/bench_detail.json $O/bench_detail.json
EF=$(python -c "import json;print(json.load(open('$O/bench.json'))['config']['ef'])" 2>/dev/null || echo 144)
echo "ef=$EF  line bytes=$(wc -c < $O/bench.json)"
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --skip-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "trace rc=$?"
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -24 $O/bench_kernel_stats.txt | cut -c1-170
rm -rf $O/trace
pmc() {  # tag, kernel regex, command...
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$?"
  done
}
pmc hnsw "hnsw_knn_kernel|distance_pairs_kernel" python $R/bench.py --skip-pagerank --skip-cpu --skip-secondary --steps 3 --warmup 1 --ef $EF
pmc pr "pb_expand_kernel|pb_reduce_kernel|pr_step_kernel" python $R/bench.py --skip-hnsw --skip-cpu --skip-secondary --pr-iters 3
pmc prrmat "pb_expand_kernel|pb_reduce_kernel|pr_hub_kernel|pr_empty_rows_kernel" python $R/scratch/r3_pr_rmat.py --kinds rmat --only-default --parity 0
pmc hnsw1m "hnsw_knn_kernel" python $R/bench.py --n 1000000 --skip-pagerank --skip-cpu --skip-secondary --steps 3 --warmup 1
pmc bfs "bfs_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py bfs 2
pmc sssp "sssp_|fill_u64_kernel" python $R/scratch/r3_rule_runs.py sssp 2
pmc cc "cc_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py cc 2
pmc tri "triangles_|tri_" python $R/scratch/r3_rule_runs.py tri 2
pmc lp "lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py lp 2
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/round4/pmc_*_*/**/*counter_collection.csv", recursive=True):
    tag = f.split("/pmc_")[1].split("/")[0]
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[tag + " " + k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/round4/pmc_summary.txt", "a") as out:
    out.write("\n# per-dispatch values in KiB as rocprofv3 reports them (uncorrected); last3avg = the timed-loop launches\n")
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            out.write(f"{k:72s} {cn:12s} n={len(vals):4d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}\n")
PY
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
# the line again with the traffic of THIS tree filled in
timeout 1500 python bench.py --skip-cpu > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err; echo "bench (traffic filled) rc=$?"
cp gpurun_out/bench_detail.json $O/bench_with_traffic_detail.json
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for f in ("bench.json", "bench_with_traffic.json"):
    d = json.load(open(R + "/gpurun_out/round4/" + f))
    print(f, "hnsw", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "ceiling", d["roofline"].get("measured_ceiling"))
    for k in ("distance_batch", "pagerank", "pagerank_rmat", "hnsw_1m", "hnsw_1m_clustered"):
        o = d.get(k, {}); print("  ", k, o.get("roofline", {}).get("frac"), o.get("roofline", {}).get("traffic"), o.get("ms_per_iteration"), o.get("inplace_reading"))
    for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
        o = d.get("graph_rules", {}).get(k, {}); print("  ", k, o.get("device_ms"), o.get("roofline", {}).get("traffic"), o.get("repeated_call_laps_ms"))
    print("   box", json.dumps(d.get("box"))[:600])
PY

#!/bin/bash
# round 6, final tree, part B: the default bench line exactly as the driver runs it (steps 20, warm-up 5; the link tables are also
# saved for the pass below), then the same workload under rocprofv3 --kernel-trace --stats on the saved index (no CPU legs)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6b
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -8
cp gpurun_out/bench_detail.json $O/bench_detail.json
echo "line bytes=$(wc -c < $O/bench.json)"
cd /tmp && export TMPDIR=/tmp
timeout 2400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --skip-cpu --index-cache /tmp/ixc > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "trace rc=$?"
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -24 $O/bench_kernel_stats.txt | cut -c1-170
rm -rf $O/trace
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for f in ("bench.json", "bench_under_rocprof.json"):
    d = json.load(open(R + "/gpurun_out/round6b/" + f))
    print(f, "hnsw", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "ceiling", d["roofline"].get("measured_ceiling"), "wall", d.get("bench_wall_s"))
    for k in ("distance_batch", "pagerank", "pagerank_rmat", "hnsw_1m", "hnsw_1m_clustered", "hnsw_10m_clustered"):
        o = d.get(k, {}); print("  ", k, o.get("value"), o.get("form"), o.get("roofline", {}).get("frac"), o.get("roofline", {}).get("traffic"), o.get("ms_per_iteration"), o.get("exact_scan"), o.get("faster_way"), o.get("skipped"))
    print("   exact_scan", d.get("exact_scan"), "ladder", d.get("batch_ladder"))
    gr = d.get("graph_rules", {})
    for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
        o = gr.get(k, {}); print("  ", k, o.get("device_ms"), o.get("roofline", {}).get("traffic"), o.get("repeated_call_laps_ms"), o.get("random_frac"))
    print("   box", json.dumps(d.get("box"))[:600])
PY

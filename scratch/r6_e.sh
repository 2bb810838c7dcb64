#!/bin/bash
# round 6: a small-size smoke of the whole bench (every new leg: in-place readings, graph rules + CPU baselines, R-MAT leg, index
# distance batch), then the distance / hnsw GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6e
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python bench.py --n 300000 --pr-nodes 1000000 --pr-edges 10000000 --steps 3 --warmup 1 > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"
grep -v Warning $O/bench_small.err | tail -25
cp gpurun_out/bench_detail.json $O/bench_small_detail.json
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = json.load(open(R + "/gpurun_out/r6e/bench_small_detail.json"))
print("line bytes", os.path.getsize(R + "/gpurun_out/r6e/bench_small.json"), "wall", d.get("bench_wall_s"))
print("distance_batch", json.dumps(d.get("distance_batch"))[:900])
pr = d.get("pagerank", {})
print("readings", json.dumps(pr.get("readings"))[:1800])
print("inplace", json.dumps({k: v for k, v in (pr.get("inplace_reading") or {}).items() if k not in ("roofline", "what")})[:1500])
for leg in ("graph_rules", "graph_rules_rmat"):
    g = d.get(leg, {})
    print(leg, g.get("graph"), g.get("error"))
    for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
        o = g.get(k, {})
        print("  ", k, "dev_ms", o.get("device_ms"), "cpu", json.dumps(o.get("cpu_baseline"))[:260], "parity", o.get("parity_checked"), o.get("parity"))
PY
timeout 900 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q -k "distance" 2>&1 | tail -3

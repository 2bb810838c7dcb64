#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3z2
timeout 900 python bench.py --skip-hnsw --skip-cpu > gpurun_out/r3z2/bench_pr_rules.json 2> gpurun_out/r3z2/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail.json"))
g = d["graph_rules"]
for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
    print(k, g[k]["device_ms"], g[k]["roofline"].get("traffic"))
print("line bytes", len(open("gpurun_out/r3z2/bench_pr_rules.json").read()))
PY

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b3
timeout 900 python bench.py --n 1000000 --skip-cpu --steps 3 --warmup 1 > gpurun_out/r3b3/bench.json 2> gpurun_out/r3b3/bench.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail.json"))
g = d["graph_rules"]
for k in ("bfs", "connected_components", "sssp"):
    print(k, g[k]["device_ms"], g[k].get("repeated_call_wall_ms"), g[k].get("repeated_call_laps_ms"))
PY

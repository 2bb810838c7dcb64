#!/bin/bash
export TMPDIR=/tmp
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import sys, time
sys.path.insert(0, ".")
sys.argv = ["bench.py"]
import bench, torch
from cozo_amd import graph as G
orig_sssp = G.sssp
def sssp(*a, **k):
    t0 = time.perf_counter()
    r = orig_sssp(*a, **k)
    print(f"sssp call {1e3 * (time.perf_counter() - t0):.1f} ms, laps {G.last_timing()}", flush=True)
    return r
G.sssp = sssp
args = bench.parse()
out = bench.bench_graph_rules(args, torch, torch.device("cuda:0"))
print({k: out[k].get("repeated_call_wall_ms") for k in ("bfs", "connected_components", "sssp")}, out.get("repeated_call_error"))
PY

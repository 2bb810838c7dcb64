#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_gpu_graph.py -q -x -k "label or lp" > gpurun_out/r3v/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3v/pytest.txt
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import sys, types, json
sys.path.insert(0, ".")
sys.argv = ["bench.py"]
import bench, torch
args = bench.parse()
out = bench.bench_graph_rules(args, torch, torch.device("cuda:0"))
for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
    e = out[k]
    print(k, {x: e[x] for x in e if x in ("device_ms", "seconds", "iterations", "colour_classes", "edges_per_s_device")})
PY

#!/bin/bash
# round 3, call i: row order of skewed PageRank plans built on the device (plan build time), overlapped exchange entry, comm tests
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_comm.py tests/test_sharded_gpu.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 600 python scratch/r3_pr_rmat.py --only-default > $O/pr.txt 2>&1; echo "pr rc=$?"; grep -E "ms/sweep|parity" $O/pr.txt
timeout 900 python bench.py --skip-hnsw > $O/bench_pr.json 2> $O/bench_pr.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_detail.json"))
for k in ("pagerank_rmat",):
    o=d.get(k,{}); print(k, o.get("ms_per_iteration"), o.get("plan_build_ms"), json.dumps(o.get("end_to_end"))[:500])
print("uniform e2e", json.dumps(d.get("end_to_end"))[:300], d.get("ms_per_step"))
g=d.get("graph_rules",{})
for k,v in g.items():
    if isinstance(v,dict): print(" ", k, v.get("device_ms"), v.get("wall_ms"), v.get("repeated_call_wall_ms"))
PY

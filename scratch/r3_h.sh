#!/bin/bash
# round 3, call h: the whole GPU suite + smoke + the default bench line on the tree as it stands (mid-round evidence), BFS A/B after 4 nodes in flight per lane group
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3h; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 600 python scratch/r3_bfs.py > $O/bfs.txt 2>&1; echo "bfs rc=$?"; grep -E "passes=|identical|held" $O/bfs.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -25
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
python - <<'PY'
import json
t=open("gpurun_out/r3h/bench.json").read()
print("line bytes", len(t))
d=json.loads(t)
print("value", d["value"], "frac", d["roofline"]["frac"], "recall", d["config"]["recall_at_k"])
for k in ("distance_batch","pagerank","pagerank_rmat","hnsw_1m","hnsw_1m_clustered"):
    o=d.get(k,{}); print(k, o.get("value"), o.get("ms_per_iteration", o.get("ms")), o.get("roofline",{}).get("frac"), o.get("parity"))
for k,v in d.get("graph_rules",{}).items():
    if isinstance(v,dict): print(" ", k, v.get("device_ms"), v.get("wall_ms"), v.get("repeated_call_wall_ms"), (v.get("roofline") or {}).get("frac"))
PY

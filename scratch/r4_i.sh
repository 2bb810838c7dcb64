#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4i
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "inplace" 2>&1 | tail -15
CZ_SSSP_TRACE=1 timeout 900 python bench.py --skip-hnsw --skip-cpu > $O/bench_pr.json 2> $O/bench_pr.err; echo "rc=$?"
grep -v "^sssp\|Warning" $O/bench_pr.err | tail -5
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4i"
d = json.load(open(O + "/bench_pr.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:400])
print("sssp", json.dumps(d.get("graph_rules", {}).get("sssp")))
PY

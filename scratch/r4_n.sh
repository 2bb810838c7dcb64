#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4n
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
CZ_PR_PLAN_TRACE=1 CZ_SSSP_TRACE=1 timeout 1500 python bench.py --skip-cpu > $O/bench.json 2> $O/bench.err; echo "rc=$?"
grep "^sssp mark" $O/bench.err | sed -n 1,40p | grep "fill dp\|sssp_run: entry"
grep -i "radix\|plan" $O/bench.err | head -12
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4n"
d = json.load(open(O + "/bench.json"))
print("hnsw", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print("sssp", json.dumps(d.get("graph_rules", {}).get("sssp")))
print("pr", d["pagerank"]["ms_per_iteration"], d["pagerank"].get("plan_build_ms"), "rmat", d["pagerank_rmat"]["ms_per_iteration"], d["pagerank_rmat"].get("plan_build_ms"))
PY

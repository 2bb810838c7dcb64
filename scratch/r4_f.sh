#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mirrors_agree.py tests/test_gpu_graph.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
CZ_SSSP_TRACE=1 timeout 1500 python bench.py --skip-cpu > $O/bench.json 2> $O/bench.err; echo "rc=$?"
grep "^sssp mark\|round 1 thr" $O/bench.err | head -80
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4f"
d = json.load(open(O + "/bench.json"))
print("hnsw", d["value"], d["ms_per_step"], json.dumps(d["roofline"]))
print("sssp", json.dumps(d.get("graph_rules", {}).get("sssp")))
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4q
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python bench.py --dist clustered --skip-pagerank --skip-cpu --skip-secondary > $O/bench_10m_clustered.json 2> $O/bench_10m_clustered.err; echo "10m rc=$?"
grep "ef sweep" $O/bench_10m_clustered.err | cut -c1-900
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4q"
d = json.load(open(O + "/bench_10m_clustered.json"))
print(d["value"], d["ms_per_step"], d["config"]["ef"], d["config"]["recall_at_k"], d["config"]["n_dist_per_query"], json.dumps(d["roofline"]))
PY

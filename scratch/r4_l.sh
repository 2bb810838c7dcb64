#!/bin/bash
# round 4, call L: is the 20-48 ms in front of the held cz_sssp_on call's first kernel a host-side WAKE-UP latency (blocked wait on the
# completion interrupt) rather than device time?  The kernel trace shows no fill_u64_kernel instance longer than a few hundred us.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4l
rm -rf $O; mkdir -p $O
cd $R
HSA_ENABLE_INTERRUPT=0 CZ_SSSP_TRACE=1 timeout 1500 python bench.py --skip-cpu > $O/bench_polling.json 2> $O/bench_polling.err; echo "rc=$?"
grep "^sssp mark\|round 1 thr 3.99" $O/bench_polling.err | head -40
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4l"
d = json.load(open(O + "/bench_polling.json"))
print("hnsw", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("measured_ceiling"))
print("sssp", json.dumps(d.get("graph_rules", {}).get("sssp")))
print("pr", d["pagerank"]["ms_per_iteration"], "dist", d["distance_batch"]["roofline"]["frac"])
PY

#!/bin/bash
# round 4, call H: ef beyond 1 024 -- BASELINE.md's 16-cluster corpus at 1M (ef 768: the ranked merge against round 3's 0.58-0.61) and at 10M
# (round 3 stopped at recall 0.78 at the ef cap); + the new GPU tests (exact_sum fallbacks, BFS claims)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "exact_sum or stale" 2>&1 | tail -4
timeout 600 python bench.py --n 1000000 --dist clustered --skip-pagerank --skip-cpu --skip-secondary > $O/bench_1m_clustered.json 2> $O/bench_1m_clustered.err; echo "1m rc=$?"
grep "ef sweep" $O/bench_1m_clustered.err | cut -c1-600
timeout 1200 python bench.py --dist clustered --skip-pagerank --skip-cpu --skip-secondary > $O/bench_10m_clustered.json 2> $O/bench_10m_clustered.err; echo "10m rc=$?"
grep "ef sweep" $O/bench_10m_clustered.err | cut -c1-700
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4h"
for f in ("bench_1m_clustered.json", "bench_10m_clustered.json"):
    try:
        d = json.load(open(O + "/" + f))
        print(f, d["value"], d["ms_per_step"], d["config"]["ef"], d["config"]["recall_at_k"], json.dumps(d["roofline"]))
    except Exception as e:
        print(f, "failed", e)
PY

#!/bin/bash
# round 3, call m: plan build with pool-backed temporaries: stage times inside bench.py, PageRank tests
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_comm.py -m gpu -x -q -k "pagerank or comm" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
CZ_PR_PLAN_TRACE=1 timeout 900 python bench.py --skip-hnsw > $O/bench.json 2> $O/bench.err; echo "rc=$?"; grep -E "\[plan\]|\[bench\] pagerank" $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print("uniform", json.dumps(d.get("end_to_end"))[:420])
print("rmat", json.dumps(d["pagerank_rmat"].get("end_to_end"))[:420])
PY

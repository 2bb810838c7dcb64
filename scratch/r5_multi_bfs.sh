#!/bin/bash
# round 5: cz_bfs_shared_until on the GPU, the forced N > 1 bench path with the per-config headline objects / exchange model
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "bfs" > $O/pytest_bfs.txt 2>&1; echo "pytest bfs rc=$?"; tail -3 $O/pytest_bfs.txt
timeout 600 python -m pytest tests/test_fixed_rule.py tests/test_mirrors_agree.py tests/test_cpp_host.py -q -m gpu > $O/pytest_rules.txt 2>&1; echo "pytest rules rc=$?"; tail -3 $O/pytest_rules.txt
CZ_BENCH_FORCE_MULTI=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --n 300000 --pr-nodes-total 4000000 --pr-edges-total 40000000 --skip-cpu > $O/bench_forced_multi.json 2> $O/bench_forced_multi.err; echo "forced multi rc=$?"; tail -3 $O/bench_forced_multi.err
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = json.load(open(R + "/gpurun_out/r5m/bench_forced_multi.json"))
print(json.dumps(d.get("headline_by_config"), indent=1))
print(json.dumps(d["pagerank"].get("exchange_model")))
PY

#!/bin/bash
# round 4: the N > 1 code of bench.py (sharded entry points, collectives, barriers, the single-process *_multi forms) with ONE rank at small
# sizes -- a smoke run of code that has never met two GPUs, not a measurement
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4o
rm -rf $O; mkdir -p $O
cd $R
CZ_BENCH_FORCE_MULTI=1 timeout 900 python bench.py --gpus 1 --n 200000 --steps 5 --warmup 2 --pr-nodes-total 2000000 --pr-edges-total 20000000 --pr-iters 5 --skip-cpu > $O/bench_forced_multi.json 2> $O/bench_forced_multi.err
echo "rc=$?"; grep -v Warning $O/bench_forced_multi.err | tail -8
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4o"
d = json.load(open(O + "/bench_forced_multi.json"))
print(len(open(O + "/bench_forced_multi.json").read()), "bytes")
print(json.dumps(d.get("single_process_multi"), indent=0)[:1800])
print(json.dumps(d.get("hnsw_sharded"))[:500])
print(json.dumps({k: d["pagerank"].get(k) for k in ("rccl_ranks_seen", "exchange_overlapped", "exchange_all_reduce", "ms_per_iteration")}))
PY
# and the launcher on a 1-GPU box: must refuse
python bench.py --gpus 2 > $O/gpus2.out 2> $O/gpus2.err; echo "--gpus 2 on this box: rc=$? stdout bytes=$(wc -c < $O/gpus2.out)"; tail -1 $O/gpus2.err

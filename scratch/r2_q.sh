#!/bin/bash
# round 2, call q: flat-pair triangles kernel (tests + time), a small default bench run (new graph_rules objects, thread ladder), and
# the N > 1 code of bench.py forced onto one rank (CZ_BENCH_FORCE_MULTI=1: sharded entry points + collectives with world = 1)
O=gpurun_out/r2q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_fixed_rule.py tests/test_zz_tie_rules.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
ONLY_ALL_SOURCES= timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules_plain.txt 2>&1
echo "rules rc=$?"; grep -v "amdgpu.ids" $O/graph_rules_plain.txt | grep -E "clustering|incidences"
timeout 900 python bench.py --n 300000 --pr-nodes 1000000 --pr-edges 10000000 --steps 5 --warmup 2 > $O/bench_small.json 2> $O/bench_small.err
echo "bench small rc=$?"; tail -3 $O/bench_small.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2q/bench_small.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','n_gpus','bench_wall_s')})
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:900])
print('graph_rules', json.dumps(d.get('graph_rules'))[:1800])
print('pagerank cpu', json.dumps(d.get('pagerank',{}).get('cpu_baseline'))[:600])
PY
CZ_BENCH_FORCE_MULTI=1 timeout 900 python bench.py --n 300000 --pr-nodes-total 1000000 --pr-edges-total 10000000 --steps 5 --warmup 2 > $O/bench_multi1.json 2> $O/bench_multi1.err
echo "bench forced-multi rc=$?"; tail -5 $O/bench_multi1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2q/bench_multi1.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','n_gpus','bench_wall_s')})
print('sharded', json.dumps(d.get('hnsw_sharded'))[:1200])
print('pagerank', json.dumps({k:v for k,v in d.get('pagerank',{}).items() if k in ('value','exchange','exchange_all_reduce','ms_per_iteration','error')})[:1200])
PY

#!/bin/bash
# round 2, call u: configs[4] (100M nodes / 1B edges) on ONE GPU: does the whole graph fit and run in 288 GB?
O=gpurun_out/r2u; mkdir -p $O
timeout 1200 python bench.py --skip-hnsw --skip-secondary --pr-nodes 100000000 --pr-edges 1000000000 --pr-iters 10 > $O/bench_pr_1b.json 2> $O/bench_pr_1b.err
echo "rc=$?"; grep -v Warning $O/bench_pr_1b.err | tail -8
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r2u/bench_pr_1b.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('metric','value','ms_per_step','bench_wall_s')}); print(json.dumps(d['roofline'])); print(json.dumps(d.get('cpu_baseline'))[:500])
    print(json.dumps(d.get('pagerank',{}).get('end_to_end'))[:800])
except Exception as e: print('no line', e)
PY
rocm-smi --showmeminfo vram 2>/dev/null | head -5

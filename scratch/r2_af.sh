#!/bin/bash
# round 2, call af: the graph-rule part of the bench line once more (closeness entry added)
O=gpurun_out/r2af; mkdir -p $O
timeout 600 python bench.py --skip-hnsw --skip-cpu > $O/bench_pr.json 2> $O/bench_pr.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2af/bench_pr.json').read().strip().splitlines()[-1])
g=d['graph_rules']
for k,v in g.items():
    if isinstance(v,dict): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if a in ('wall_ms','device_ms','repeated_call_wall_ms','error','iterations')})
    else: print(k, str(v)[:100])
PY

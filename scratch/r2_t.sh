#!/bin/bash
# round 2, final call: the whole GPU suite + smoke + the default bench line on the final tree
O=gpurun_out/r2final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -12
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2final/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','bench_wall_s')}, d['roofline']['frac'], d['roofline']['traffic'])
print('cpu', json.dumps(d['cpu_baseline'])[:700])
print('dist', d['distance_batch']['roofline']['frac'], 'pr', d['pagerank']['roofline']['frac'], d['pagerank']['ms_per_iteration'])
for k,v in d['graph_rules'].items():
    if isinstance(v, dict): print(k, {a:b for a,b in v.items() if a in ('wall_ms','device_ms','edges_per_s_device','iterations','colour_classes')})
PY

#!/bin/bash
# round 2, call w: mirrors differential test on the device; bench N > 1 code forced onto one rank again (after the fallback edit)
O=gpurun_out/r2w; mkdir -p $O
timeout 900 python -m pytest tests/test_mirrors_agree.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
CZ_BENCH_FORCE_MULTI=1 timeout 900 python bench.py --n 300000 --pr-nodes-total 1000000 --pr-edges-total 10000000 --steps 5 --warmup 2 > $O/bench_multi1.json 2> $O/bench_multi1.err
echo "bench forced-multi rc=$?"; tail -3 $O/bench_multi1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2w/bench_multi1.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','n_gpus','bench_wall_s')}); print(d['pagerank']['exchange'][:120]); print(json.dumps(d.get('hnsw_sharded'))[:300])
PY

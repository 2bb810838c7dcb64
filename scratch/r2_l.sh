#!/bin/bash
# round 2, call l: PageRank tile 64 KiB vs 128 KiB; SSSP bucket width
O=gpurun_out/r2l; mkdir -p $O
for v in base tile15; do
  if [ $v = base ]; then unset COZO_GPU_LIB; else export COZO_GPU_LIB=$PWD/scratch/lib/libcozo_gpu_$v.so; fi
  timeout 600 python bench.py --skip-hnsw --skip-cpu > $O/pr_$v.json 2> $O/pr_$v.err; echo "pr $v rc=$?"
done
unset COZO_GPU_LIB
python - <<'PY'
import json
for v in ("base", "tile15"):
    d = json.load(open(f"gpurun_out/r2l/pr_{v}.json"))
    r = d["pagerank_rmat"]
    print(v, "uniform", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 4), "| rmat exact", round(r["roofline"]["avg_launch_ms"], 4),
          "| relaxed", round(r.get("relaxed", {}).get("roofline", {}).get("avg_launch_ms", 0), 4), "| graph_rules", {k: round(x["wall_ms"], 1) for k, x in d.get("graph_rules", {}).items() if isinstance(x, dict)})
PY
for dl in 1 2 8 inf; do
  CZ_SSSP_DELTA=$dl GN=10000000 timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from cozo_amd import _lib
L = _lib.lib(); assert L.cz_init(0) == 0
import numpy as np, torch
from cozo_amd import graph as G
dev = torch.device("cuda:0")
n, e = 10_000_000, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
keep = src != dst
key = torch.unique(src[keep] * n + dst[keep])
s = torch.div(key, n, rounding_mode="floor"); t = key - s * n
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
ooff, otgt = off.to(torch.int32).cpu().numpy().view(np.uint32), t.to(torch.int32).cpu().numpy().view(np.uint32)
w = (torch.randint(1, 64, (otgt.size,), generator=g, device=dev, dtype=torch.int32).to(torch.float32) / 8).cpu().numpy()
starts = np.array([0], dtype=np.uint32)
G.sssp(ooff, otgt, w, starts)
t0 = time.perf_counter(); G.sssp(ooff, otgt, w, starts); dt = time.perf_counter() - t0
print(f"CZ_SSSP_DELTA={os.environ['CZ_SSSP_DELTA']}: cz_sssp wall {dt * 1e3:.1f} ms", flush=True)
PY
done

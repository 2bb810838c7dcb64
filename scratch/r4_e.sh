#!/bin/bash
# round 4, call E: the whole bench (HBM probe + box telemetry in the line) with the SSSP trace on: cold call against the held call
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e
rm -rf $O; mkdir -p $O
cd $R
CZ_SSSP_TRACE=1 timeout 1500 python bench.py --skip-cpu > $O/bench.json 2> $O/bench.err; echo "rc=$?"
cp gpurun_out/bench_detail.json $O/bench_detail.json
grep -v "Warning\|^sssp phase" $O/bench.err | tail -8
python3 - <<'PY'
import json, os, re
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4e"
d = json.load(open(O + "/bench.json"))
print("hnsw", d["value"], d["ms_per_step"], json.dumps(d["roofline"]))
print("dist", json.dumps(d.get("distance_batch")))
print("1m", json.dumps(d.get("hnsw_1m", {}).get("roofline")))
print("box", json.dumps(d.get("box")))
print("sssp", json.dumps(d.get("graph_rules", {}).get("sssp")))
# the trace: split into calls at "round 1"
calls, cur = [], None
for line in open(O + "/bench.err"):
    m = re.match(r"sssp phase (\d+) round (\d+) thr (\S+) near (\d+) far (\d+)  \(\+([\d.]+) us", line)
    if not m: continue
    ph, rd, thr, near, far, us = int(m[1]), int(m[2]), float(m[3]), int(m[4]), int(m[5]), float(m[6])
    if rd == 1:
        cur = []; calls.append(cur)
    cur.append((ph, rd, near, far, us))
for i, c in enumerate(calls[:8]):
    tot = sum(x[4] for x in c[1:])
    print(f"call {i}: {len(c)} relax rounds, entries {sum(x[2] for x in c)}, time between first and last line {tot / 1e3:.2f} ms, slowest rounds:",
          sorted(((round(x[4]), x[1], x[2]) for x in c[1:]), reverse=True)[:4])
PY

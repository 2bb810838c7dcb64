#!/bin/bash
# the whole bench with the SSSP schedule traced: is the slow held call a different schedule or the same schedule running slower?
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b4
CZ_SSSP_TRACE=1 timeout 1200 python bench.py --skip-cpu > gpurun_out/r3b4/bench.json 2> gpurun_out/r3b4/bench.err; echo "rc=$?"
python - <<'PY'
import json, re
calls, cur = [], None
for line in open("gpurun_out/r3b4/bench.err"):
    m = re.match(r"sssp phase (\d+) round (\d+) thr (\S+) near (\d+) far (\d+)", line)
    if not m:
        continue
    ph, rd, thr, near, far = int(m[1]), int(m[2]), float(m[3]), int(m[4]), int(m[5])
    if ph == 1 and rd == 1:
        cur = {"rounds": 0, "entries": 0, "first_thr": thr, "max_near": 0}
        calls.append(cur)
    cur["rounds"] += 1
    cur["entries"] += near
    cur["max_near"] = max(cur["max_near"], near)
print(len(calls), "SSSP runs traced; the first eight:")
for c in calls[:8]:
    print("  ", c)
d = json.load(open("gpurun_out/bench_detail.json"))
g = d["graph_rules"]["sssp"]
print("sssp device", g["device_ms"], "held wall", g.get("repeated_call_wall_ms"), "held laps", g.get("repeated_call_laps_ms"))
PY

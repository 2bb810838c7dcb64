#!/bin/bash
# round 6: in-place plan after the branch-free loads: resident-plan tests, per-level trace, HBM traffic counters of one sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "inplace or vouched" > $O/pytest_inplace.txt 2>&1; echo "pytest inplace rc=$?"; tail -3 $O/pytest_inplace.txt
cd /tmp && export TMPDIR=/tmp
CFG=${TRACE_CFG:-t16s32}
IP_CFGS=$CFG IP_PARITY=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o ip -- python $R/scratch/r6_inplace.py uniform > $O/trace_run.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python - "$db" <<'PY' > $O/inplace_levels.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x from kernels where name like '%gi_%' order by start").fetchall()
ends = [i for i, r in enumerate(rows) if 'sum_partials' in r[0]]
a, b = ends[-2] + 1, ends[-1]
t0 = rows[a][1]
prev = t0
for name, st, en, g in rows[a:b + 1]:
    short = name.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[-22:]
    print(f"{short:22s} wgs {g // 1024:6d} start {(st - t0) / 1e3:9.2f} us  gap {(st - prev) / 1e3:6.2f}  dur {(en - st) / 1e3:8.2f} us")
    prev = en
print("sweep span us", (rows[b][2] - t0) / 1e3)
PY
tail -42 $O/inplace_levels.txt
rm -rf $O/trace
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  IP_CFGS=$CFG IP_PARITY=0 IP_FEW=1 timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o pmc -- python $R/scratch/r6_inplace.py uniform > $O/p$i.out 2>&1
  echo "pass $i ($set) rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/r6c/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/r6c/pmc_summary.txt", "w") as out:
    for k in sorted(acc):
        if "gi_" not in k:
            continue
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:30s} {cn:24s} n={len(vals):5d} sum={sum(vals):.6g} avg={sum(vals)/len(vals):.4g}"
            print(line); out.write(line + "\n")
PY
rm -rf $O/p[0-9]

#!/bin/bash
# round 6: the fused level launches of the in-place plan: tests, tile / slice / gap sweep, kernel trace of one setting
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "inplace or vouched" > $O/pytest_inplace.txt 2>&1; echo "pytest inplace rc=$?"; tail -5 $O/pytest_inplace.txt
timeout 1500 python scratch/r6_inplace.py uniform > $O/inplace_uniform.txt 2>&1; echo "probe rc=$?"; grep -v Warning $O/inplace_uniform.txt | cut -c1-330 | tail -12
cd /tmp && export TMPDIR=/tmp
IP_CFGS=${TRACE_CFG:-t16s16} IP_PARITY=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o ip -- python $R/scratch/r6_inplace.py uniform > $O/trace_run.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/inplace_kernel_stats.txt; head -8 $O/inplace_kernel_stats.txt | cut -c1-150
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
tail -45 $O/inplace_levels.txt
rm -rf $O/trace

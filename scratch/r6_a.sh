#!/bin/bash
# round 6, first GPU call: the GPU suite on the tree with the ADVICE fixes + the resident in-place plan, then the plan on the bench graph
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6a
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "inplace or vouched" > $O/pytest_inplace.txt 2>&1; echo "pytest inplace rc=$?"; tail -5 $O/pytest_inplace.txt
timeout 900 python scratch/r6_inplace.py uniform > $O/inplace_uniform.txt 2>&1; echo "probe rc=$?"; grep -v Warning $O/inplace_uniform.txt | tail -12
cd /tmp && export TMPDIR=/tmp
IP_CFGS=gap1 IP_PARITY=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o ip -- python $R/scratch/r6_inplace.py uniform > $O/trace_run.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/inplace_kernel_stats.txt; head -14 $O/inplace_kernel_stats.txt | cut -c1-150
python - "$db" <<'PY' > $O/inplace_levels.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = c.execute("select name, start, end, grid_x from kernels where name like '%gi_%' order by start").fetchall()
# the last full sweep: from the last-but-one gi_sum_partials to the last
ends = [i for i, r in enumerate(rows) if 'sum_partials' in r[0]]
a, b = ends[-2] + 1, ends[-1]
t0 = rows[a][1]
for name, st, en, g in rows[a:b + 1]:
    short = name.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[-22:]
    print(f"{short:22s} grid {g:8d} start {(st - t0) / 1e3:9.2f} us  dur {(en - st) / 1e3:8.2f} us")
print("sweep span us", (rows[b][2] - t0) / 1e3)
PY
tail -5 $O/inplace_levels.txt
rm -rf $O/trace
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"; tail -5 $O/pytest_gpu.txt

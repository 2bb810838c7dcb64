#!/bin/bash
# round 5, call e: accumulate formulation with plan-time piece lists -- parity tests, kernel trace on the uniform graph, both graphs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "pagerank" > $O/pytest_pr.txt 2>&1
tail -5 $O/pytest_pr.txt
cd /tmp && export TMPDIR=/tmp
PR_CFGS=blocked,acc,acc_b2,acc_w16 rocprofv3 --kernel-trace --stats -d $O/prof -o pr -- python $R/scratch/r5_pr.py both > $O/prof_out.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python - "$db" <<'PY' > $O/kernels_by_shape.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for pat in ('%pb_expand%', '%pb_reduce%', '%pa_reduce%', '%pr_hub%'):
    for r in c.execute("select grid_x, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels where name like ? group by grid_x, lds_size order by grid_x desc", (pat,)):
        print(pat, "wgs", r[0] // r[1], "threads", r[1], "lds", r[2], "n", r[3], "avg_us", round(r[4] / 1e3, 1), "min", round(r[5] / 1e3, 1), "max", round(r[6] / 1e3, 1))
PY
cat $O/kernels_by_shape.txt
rm -rf $O/prof
grep -v "^/opt" $O/prof_out.txt | cut -c1-150

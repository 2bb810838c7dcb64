#!/bin/bash
# rocprofv3 kernel trace of the pagerank sweep script
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_pr
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pr -o pr -- python $R/scratch/pr_sweep.py "$@" > $R/gpurun_out/prof_pr/out.txt 2>&1
echo "rc=$?"
db=$(find $R/gpurun_out/prof_pr -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $R/gpurun_out/prof_pr/kernel_stats.txt
grep -E "pb_|pr_|^# total|^kernel" $R/gpurun_out/prof_pr/kernel_stats.txt | cut -c1-190
tail -8 $R/gpurun_out/prof_pr/out.txt
python - <<'PY'
import sqlite3, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
db = glob.glob(R + "/gpurun_out/prof_pr/*.db")[0]
c = sqlite3.connect(db)
for pat in ('%pb_expand%', '%pb_reduce%', '%pr_step%'):
    for r in c.execute("select grid_x, workgroup_x, count(*), avg(duration), min(duration), max(duration) from kernels where name like ? group by grid_x order by grid_x desc", (pat,)):
        print(pat, "grid", r[0] // r[1], "n", r[2], "avg_us", round(r[3] / 1e3, 1), "min", round(r[4] / 1e3, 1), "max", round(r[5] / 1e3, 1))
PY

#!/bin/bash
# round 5, call c: where the accumulate sweep's time goes (kernel trace), and the build-time variants of phase B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PR_CFGS=blocked,acc,acc_b2,acc_a2b2 rocprofv3 --kernel-trace --stats -d $O/prof -o pr -- python $R/scratch/r5_pr.py uniform > $O/prof_out.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python - "$db" <<'PY' > $O/kernels_by_shape.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for pat in ('%pb_expand%', '%pb_reduce%', '%pa_reduce%'):
    for r in c.execute("select grid_x, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels where name like ? group by grid_x, lds_size order by grid_x desc", (pat,)):
        print(pat, "wgs", r[0] // r[1], "threads", r[1], "lds", r[2], "n", r[3], "avg_us", round(r[4] / 1e3, 1), "min", round(r[5] / 1e3, 1), "max", round(r[6] / 1e3, 1))
PY
cat $O/kernels_by_shape.txt
rm -rf $O/prof
cd $R
for v in u4 u16 nt0; do
  echo "== variant $v" >> $O/variants.txt
  COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_$v.so PR_CFGS=acc,acc_b2,acc_a2b2 python scratch/r5_pr.py uniform >> $O/variants.txt 2>&1
done
cat $O/variants.txt

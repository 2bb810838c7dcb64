#!/bin/bash
# round 5, call d: ablations of pa_reduce_kernel (which part of a piece costs the time), kernel trace per variant
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in main abl1 abl2 abl3; do
  lib=$R/scratch/lib/libcozo_gpu_$v.so; [ $v = main ] && lib=$R/cozo_amd/lib/libcozo_gpu.so
  COZO_GPU_LIB=$lib PR_CFGS=acc,acc_b2 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o pr -- python $R/scratch/r5_pr.py uniform > $O/out_$v.txt 2>&1
  db=$(find $O/prof_$v -name "*.db" | head -1)
  echo "== $v" >> $O/kernels.txt
  python - "$db" <<'PY' >> $O/kernels.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for pat in ('%pb_expand%', '%pa_reduce%'):
    for r in c.execute("select grid_x, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels where name like ? group by grid_x, lds_size order by grid_x desc", (pat,)):
        print(pat, "wgs", r[0] // r[1], "threads", r[1], "lds", r[2], "n", r[3], "avg_us", round(r[4] / 1e3, 1), "min", round(r[5] / 1e3, 1), "max", round(r[6] / 1e3, 1))
PY
  rm -rf $O/prof_$v
done
cat $O/kernels.txt

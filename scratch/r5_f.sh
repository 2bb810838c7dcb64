#!/bin/bash
# round 5, call f: PMC passes over the accumulate sweep (uniform graph): HBM bytes, wave states, LDS conflicts, L2 hits; NT on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PR_CFGS=acc,acc_w16
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o pmc -- python $R/scratch/r5_pr.py uniform > $O/p$i.out 2>&1
  echo "pass $i ($set) rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/r5f/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] + " grid=" + row.get("Grid_Size", "?") + " wg=" + row.get("Workgroup_Size", "?")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/r5f/summary.txt", "w") as out:
    for k in sorted(acc):
        if not any(t in k for t in ("pb_expand", "pb_reduce", "pa_reduce")):
            continue
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:70s} {cn:28s} n={len(vals):3d} avg={sum(vals)/len(vals):.5g}"
            print(line); out.write(line + "\n")
PY
cd $R
echo "== NT=0 variant" > $O/nt0.txt
COZO_GPU_LIB=$R/scratch/lib/libcozo_gpu_nt0.so PR_CFGS=acc,acc_w16,acc_b2 python scratch/r5_pr.py uniform >> $O/nt0.txt 2>&1
grep -v "^/opt" $O/nt0.txt | cut -c1-120

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFG=${TRACE_CFG:-t16s32}
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

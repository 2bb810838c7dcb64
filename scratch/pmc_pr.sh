#!/bin/bash
# PMC passes (each its own run, no tracing) over the pagerank sweep script; summarised per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_pr
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PR_SWEEP=blocked_only
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o pmc -- python $R/scratch/pr_sweep.py "$@" > $O/p$i.out 2>&1
  echo "pass $i ($set) rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_pr/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/pmc_pr/summary.txt", "w") as out:
    for k in sorted(acc):
        if not any(t in k for t in ("pb_expand", "pb_reduce", "pr_step")):
            continue
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:40s} {cn:24s} n={len(vals):3d} avg={sum(vals)/len(vals):.4g}"
            print(line); out.write(line + "\n")
PY

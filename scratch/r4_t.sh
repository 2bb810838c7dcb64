#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4t
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python scratch/r4_pr_runs.py 2>&1 | grep "slice" | tee $O/times.txt
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "pb_expand_kernel|pb_reduce_kernel" --output-format csv -d $O/pmc_$set -o pmc -- python $R/scratch/r4_pr_runs.py > $O/pmc_$set.out 2>&1; echo "pmc $set rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4t"
for set_, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    per = collections.defaultdict(list)
    for f in glob.glob(O + f"/pmc_{set_}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = "A pb_expand" if "pb_expand" in r["Kernel_Name"] else "B pb_reduce"
            per[k].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    for k in sorted(per):
        v = sorted(per[k])
        # consecutive dispatches with the same grid = one slice setting
        groups, cur = [], []
        for d, val, grid in v:
            if cur and grid != cur[-1][2]:
                groups.append(cur); cur = []
            cur.append((d, val, grid))
        if cur: groups.append(cur)
        print(set_, "phase", k, "| " + " | ".join(f"grid {g[0][2]}: {len(g)} launches, {sum(x[1] for x in g) / len(g) * 1024 * scale / 1e6:7.1f} MB each" for g in groups))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE

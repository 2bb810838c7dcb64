#!/bin/bash
# round 5: the translation counters of hnsw_knn_kernel per handle landing (scratch/r5_landing_pmc.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5lpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "${PMC_SET:-TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum}" ; do
  timeout 900 rocprofv3 --pmc $set --kernel-include-regex "hnsw_knn_kernel" --output-format csv -d $O/pmc -o pmc -- python $R/scratch/r5_landing_pmc.py > $O/run.out 2>&1
  echo "rc=$?"
done
grep HANDLE $O/run.out
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
O = R + "/gpurun_out/r5lpmc"
rows = []
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
print("dispatches", len(ids))
handles = [l.split() for l in open(O + "/run.out") if l.startswith("HANDLE")]
with open(O + "/summary.txt", "w") as out:
    for h, tag in enumerate(handles):
        chunk = ids[h * 28 + 4:(h + 1) * 28]
        if not chunk: break
        names = sorted({k for i in chunk for k in by[i]})
        line = " ".join(tag) + "   " + "   ".join(f"{nm} / launch {sum(by[i].get(nm, 0) for i in chunk) / len(chunk):.5g}" for nm in names)
        print(line); out.write(line + "\n")
PY
rm -rf $O/pmc

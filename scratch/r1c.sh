#!/bin/bash
# round-1 (session c) measurement call: gpu tests, bench under rocprofv3 kernel trace, hnsw PMC pass, sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "trace rc=$?"; tail -2 $O/bench_under_rocprof.err; cat $O/bench_under_rocprof.json
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -12 $O/bench_kernel_stats.txt | cut -c1-160
rm -rf $O/trace
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "hnsw_knn_kernel" --output-format csv -d $O/pmc_$set -o pmc -- python $R/bench.py --skip-cpu --skip-pagerank --steps 3 --warmup 1 --ef 96 > $O/pmc_$set.out 2>&1
  echo "pmc $set rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/r1c/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/r1c/hnsw_pmc_summary.txt", "w") as out:
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:60s} {cn:16s} n={len(vals):3d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}"
            print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cd $R
HS_US=2,4,8 timeout 600 python scratch/hnsw_sweep.py > $O/hnsw_sweep.txt 2>&1; echo "hnsw_sweep rc=$?"; cat $O/hnsw_sweep.txt | tail -20
PR_SWEEP=chunks timeout 600 python scratch/pr_sweep.py > $O/pr_sweep.txt 2>&1; echo "pr_sweep rc=$?"; cat $O/pr_sweep.txt | tail -12

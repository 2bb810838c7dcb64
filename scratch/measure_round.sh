#!/bin/bash
# one round's measurement set: plain bench, bench under rocprofv3 kernel trace, PMC traffic passes (copy the summaries to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "trace rc=$?"
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -14 $O/bench_kernel_stats.txt | cut -c1-150; grep "^# " $O/bench_kernel_stats.txt | tail -8
rm -rf $O/trace
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "hnsw_knn_kernel|distance_pairs_kernel|distance_runs_kernel|pb_expand_kernel|pb_reduce_kernel" --output-format csv -d $O/pmc_$set -o pmc -- python $R/bench.py --skip-cpu --steps 3 --warmup 1 --ef 96 --pr-iters 3 > $O/pmc_$set.out 2>&1
  echo "pmc $set rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/round/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/round/pmc_summary.txt", "w") as out:
    out.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs, no tracing) over `python bench.py --skip-cpu --steps 3 --warmup 1 --ef 96 --pr-iters 3`\n# per-dispatch values in KiB as rocprofv3 reports them (uncorrected); last3avg = the timed-loop launches\n")
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:56s} {cn:12s} n={len(vals):3d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}"
            print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# end to end on a stored relation: bytes -> libcozo_ingest (host) -> cz_pagerank on host arrays (upload + plan + run + scores back)
cd $R && timeout 900 python scratch/e2e_pagerank_stored.py --rows 100000000 --nodes 10000000 > $O/e2e_pagerank_stored.txt 2>&1; echo "e2e rc=$?"; cat $O/e2e_pagerank_stored.txt

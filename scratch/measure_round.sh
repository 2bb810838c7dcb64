#!/bin/bash
# one round's measurement set: plain bench, bench under rocprofv3 kernel trace, PMC traffic passes (copy the summaries to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_stored_index_cpp_gpu.py tests/test_gpu_graph.py tests/test_gpu_comm.py tests/test_cpp_host.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; echo "pytest subset rc=$?"; tail -4 $O/pytest_subset.txt
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -40
EF=$(python -c "import json;print(json.load(open('$O/bench.json'))['config']['ef'])" 2>/dev/null || echo 144)
echo "ef=$EF"
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --skip-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "trace rc=$?"
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -30 $O/bench_kernel_stats.txt | cut -c1-170
rm -rf $O/trace
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $set --kernel-include-regex "hnsw_knn_kernel|distance_pairs_kernel" --output-format csv -d $O/pmch_$set -o pmc -- python $R/bench.py --skip-pagerank --skip-cpu --skip-secondary --steps 3 --warmup 1 --ef $EF > $O/pmch_$set.out 2>&1
  echo "pmc hnsw $set rc=$?"
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "pb_expand_kernel|pb_reduce_kernel|pr_step_kernel|pr_hub_finish_kernel" --output-format csv -d $O/pmcp_$set -o pmc -- python $R/bench.py --skip-hnsw --skip-cpu --skip-secondary --pr-iters 3 > $O/pmcp_$set.out 2>&1
  echo "pmc pagerank $set rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/round/pmc*_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/round/pmc_summary.txt", "w") as out:
    out.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs, no tracing) over `python bench.py --skip-pagerank --skip-cpu --skip-secondary --steps 3 --warmup 1 --ef EF` (HNSW 10M x 768 + distance batch) and `python bench.py --skip-hnsw --skip-cpu --skip-secondary --pr-iters 3` (PageRank uniform)\n# per-dispatch values in KiB as rocprofv3 reports them (uncorrected); last3avg = the timed-loop launches\n")
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:56s} {cn:12s} n={len(vals):3d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}"
            print(line); out.write(line + "\n")
PY
rm -rf $O/pmch_FETCH_SIZE $O/pmch_WRITE_SIZE $O/pmcp_FETCH_SIZE $O/pmcp_WRITE_SIZE

#!/bin/bash
# round 4, final tree, part A: the PMC traffic passes again for everything whose device sources changed since the last set
# (hnsw_kernels.cuh: merge_wide; graph.hip: the sharded SSSP backend, the BFS claim pre-read), with ONE build of the 10M index
# shared by the passes (bench.py --index-cache), then the GPU suite.  PageRank's entries keep their hashes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round4c
rm -rf $O; mkdir -p $O
cp $R/profiles/r04_bench_final_detail.json $O/bench_detail.json   # (algorithmic bytes of the workloads for make_pmc_traffic.py)
EF=$(python -c "import json;print(json.load(open('$R/profiles/r04_bench_final.json'))['config']['ef'])" 2>/dev/null || echo 144)
echo "ef=$EF"
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag, kernel regex, command...
  local tag=$1 rx=$2; shift 2
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${tag}_$set -o pmc -- "$@" > $O/pmc_${tag}_$set.out 2>&1
    echo "pmc $tag $set rc=$? ($(date +%T))"
  done
}
pmc hnsw "hnsw_knn_kernel|distance_pairs_kernel" python $R/bench.py --skip-pagerank --skip-cpu --skip-secondary --steps 3 --warmup 1 --ef $EF --index-cache /tmp/ixc
pmc hnsw1m "hnsw_knn_kernel" python $R/bench.py --n 1000000 --skip-pagerank --skip-cpu --skip-secondary --steps 3 --warmup 1 --index-cache /tmp/ixc
pmc bfs "bfs_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py bfs 2
pmc sssp "sssp_|fill_u64_kernel" python $R/scratch/r3_rule_runs.py sssp 2
pmc cc "cc_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py cc 2
pmc tri "triangles_|tri_" python $R/scratch/r3_rule_runs.py tri 2
pmc lp "lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py lp 2
grep -h "index\|Traceback\|Error" $O/pmc_hnsw_*.out | grep -v Warning | head -8
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/round4c/pmc_*_*/**/*counter_collection.csv", recursive=True):
    tag = f.split("/pmc_")[1].split("/")[0]
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[tag + " " + k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/round4c/pmc_summary.txt", "a") as out:
    out.write("\n# per-dispatch values in KiB as rocprofv3 reports them (uncorrected); last3avg = the timed-loop launches\n")
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            out.write(f"{k:72s} {cn:12s} n={len(vals):4d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}\n")
PY
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.txt

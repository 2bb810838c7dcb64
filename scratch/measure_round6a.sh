#!/bin/bash
# round 6, final tree, part A: the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs, no tracing) for every entry of
# profiles/pmc_traffic.json -- the device sources of all of them changed this round -- with ONE build of the 10M index shared by
# the passes (bench.py --index-cache), an MFMA-utilisation pass over the exhaustive scan, then the GPU suite.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6a
rm -rf $O; mkdir -p $O
cd $R
# the workloads' algorithmic bytes (make_pmc_traffic.py reads them from a bench detail file of the same tree): a run without CPU legs
timeout 1200 python bench.py --skip-cpu --skip-clustered-10m --index-cache /tmp/ixc > $O/bench_nocpu.json 2> $O/bench_nocpu.err; echo "bench (no cpu legs) rc=$? ($(date +%T))"
grep -v Warning $O/bench_nocpu.err | tail -4
cp gpurun_out/bench_detail.json $O/bench_detail.json
EF=$(python -c "import json;print(json.load(open('$O/bench_nocpu.json'))['config']['ef'])" 2>/dev/null || echo 144)
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
pmc pr "pb_expand_kernel|pb_reduce_kernel|pa_reduce_kernel|pr_hub_kernel|pr_empty_rows_kernel" python $R/bench.py --skip-hnsw --skip-cpu --skip-secondary --pr-iters 3
IP_CFGS=t16s16 IP_PARITY=0 IP_FEW=1 pmc prip "gi_level_kernel|gi_long_kernel|gi_sum_partials_kernel" python $R/scratch/r6_inplace.py uniform
pmc prrmat "pb_expand_kernel|pb_reduce_kernel|pa_reduce_kernel|pr_hub_kernel|pr_empty_rows_kernel" python $R/scratch/r5_pr_rmat_runs.py 3
pmc bfs "bfs_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py bfs 2
pmc sssp "sssp_|fill_u64_kernel" python $R/scratch/r3_rule_runs.py sssp 2
pmc cc "cc_|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py cc 2
pmc tri "triangles_|tri_" python $R/scratch/r3_rule_runs.py tri 2
pmc lp "lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel" python $R/scratch/r3_rule_runs.py lp 2
# the matrix cores during the exhaustive scan (the ground-truth leg of a 300k bench: dot_gemm_mfma_kernel over 300k x 768 x 1024 queries)
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_MFMA"; do
  tag=mfma_$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "dot_gemm_mfma_kernel" --output-format csv -d $O/pmc_$tag -o pmc -- python $R/bench.py --n 300000 --skip-pagerank --skip-cpu --skip-secondary --steps 2 --warmup 1 > $O/pmc_$tag.out 2>&1; echo "pmc $tag rc=$?"
done
grep -h "Traceback\|Error" $O/pmc_*.out | grep -v Warning | head -8
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/round6a/pmc_*/**/*counter_collection.csv", recursive=True):
    tag = f.split("/pmc_")[1].split("/")[0]
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[tag + " " + k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R + "/gpurun_out/round6a/pmc_summary.txt", "a") as out:
    out.write("\n# per-dispatch values as rocprofv3 reports them (FETCH_SIZE / WRITE_SIZE in KiB, uncorrected); last3avg = the timed-loop launches\n")
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            out.write(f"{k:72s} {cn:28s} n={len(vals):4d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}\n")
PY
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.txt

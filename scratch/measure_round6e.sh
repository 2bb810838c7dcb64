#!/bin/bash
# round 6, part E: after a change to hnsw_kernels.h / hnsw_api.hip only -- the HNSW GPU tests, then the PMC passes of the three entries
# tied to those sources (hnsw_knn, hnsw_knn_1m, distance_batch) with ONE build of the 10M index, then the large-ef comparison.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6e
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_hnsw_build.py -q -x > $O/pytest_hnsw.txt 2>&1; echo "pytest hnsw rc=$?"; tail -2 $O/pytest_hnsw.txt
timeout 1200 python bench.py --skip-cpu --skip-clustered-10m --skip-pagerank --skip-secondary --index-cache /tmp/ixc > $O/bench_nocpu.json 2> $O/bench_nocpu.err; echo "bench (hnsw only) rc=$? ($(date +%T))"
grep -v Warning $O/bench_nocpu.err | tail -4
cp $R/profiles/r06_pmc_bench_detail.json $O/bench_detail.json
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
cd $R
python profiles/make_pmc_traffic.py $O > $O/pmc_summary.txt 2>&1; grep -i "hnsw\|distance" $O/pmc_summary.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
timeout 900 python scratch/r6_pending.py 2>&1 | grep -v Warning | tail -9 > $O/pending_1m.txt; cat $O/pending_1m.txt

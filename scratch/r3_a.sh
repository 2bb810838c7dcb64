#!/bin/bash
# round 3, call a: exact parallel row sums (exact_sum.cuh) -- PageRank parity tests, then uniform + R-MAT sweeps with the kernel trace
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
cd $R
tests/cpp/bin/exact_sum_test > $O/exact_sum_host.txt 2>&1; echo "host exact-sum rc=$?"
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "pagerank" > $O/pytest_pagerank.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pagerank.txt
timeout 900 python bench.py --skip-hnsw > $O/bench_pr.json 2> $O/bench_pr.err; echo "bench rc=$?"; grep -v Warning $O/bench_pr.err | tail -20
cp gpurun_out/bench_detail.json $O/bench_pr_detail.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a/bench_pr.json"))
for k in ("pagerank_rmat",):
    r=d.get(k,{}); print(k, r.get("ms_per_iteration"), r.get("roofline"), r.get("parity"))
print("uniform", d.get("ms_per_step"), d.get("roofline"))
print("line bytes", len(open("gpurun_out/r3a/bench_pr.json").read()))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o pr -- python $R/bench.py --skip-hnsw --skip-cpu > $R/$O/bench_pr_traced.json 2> $R/$O/bench_pr_traced.err
echo "trace rc=$?"
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; head -40 $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

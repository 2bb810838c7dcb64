#!/bin/bash
# round 2, call ad: PageRank with the long rows on a side stream (R-MAT exact mode), parity tests, PMC passes for the uniform sweep
R=$GRAFT_REPO_ROOT; O=gpurun_out/r2ad; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_comm.py -m gpu -q -k "pagerank or comm or child" > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 900 python bench.py --skip-hnsw --skip-cpu > $O/bench_pr.json 2> $O/bench_pr.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2ad/bench_pr.json').read().strip().splitlines()[-1])
print('uniform', d['value'], d['ms_per_step'], d['roofline']['frac'])
r=d.get('pagerank_rmat',{}); print('rmat', r.get('ms_per_iteration'), r.get('parity'), 'relaxed', r.get('relaxed',{}).get('ms_per_iteration'))
PY
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "pb_expand_kernel|pb_reduce_kernel|pr_step_kernel|pr_hub_finish_kernel" --output-format csv -d $R/$O/pmcp_$set -o pmc -- python $R/bench.py --skip-hnsw --skip-cpu --skip-secondary --pr-iters 3 > $R/$O/pmcp_$set.out 2>&1
  echo "pmc pagerank $set rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r2ad/pmcp_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/r2ad/pmc_summary.txt", "w") as out:
    for k in sorted(acc):
        for cn, vals in sorted(acc[k].items()):
            line = f"{k:40s} {cn:12s} n={len(vals):3d} avg={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g} last3avg={sum(vals[-3:])/len(vals[-3:]):.6g}"
            print(line); out.write(line + "\n")
PY
rm -rf $O/pmcp_FETCH_SIZE $O/pmcp_WRITE_SIZE

#!/bin/bash
# round 2, call d: comm tests at N = 1, distance-batch microbench, PageRank on uniform + R-MAT (exact / relaxed / flat) with a kernel trace
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_graph.py tests/test_gpu_hnsw.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 600 python scratch/r2_dist.py > $O/dist.txt 2>&1; echo "dist rc=$?"; cat $O/dist.txt | grep -v amdgpu.ids
timeout 600 python bench.py --skip-hnsw --skip-cpu > $O/pr.json 2> $O/pr.err; echo "pr rc=$?"
CZ_PR_FLAT=1 timeout 600 python bench.py --skip-hnsw --skip-cpu > $O/pr_flat.json 2> $O/pr_flat.err; echo "pr_flat rc=$?"
python - <<'PY'
import json
for f in ("pr", "pr_flat"):
    d = json.load(open(f"gpurun_out/r2d/{f}.json"))
    r = d["pagerank_rmat"]
    print(f, "uniform", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "| rmat exact", r["ms_per_iteration"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"],
          "| relaxed", r.get("relaxed", {}).get("ms_per_iteration"), r.get("relaxed", {}).get("roofline", {}).get("avg_launch_ms"), r.get("relaxed", {}).get("error"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o pr -- python $GRAFT_REPO_ROOT/bench.py --skip-hnsw --skip-cpu > $GRAFT_REPO_ROOT/$O/pr_prof.json 2> $GRAFT_REPO_ROOT/$O/pr_prof.err
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/pr_kernel_stats.txt; head -20 $O/pr_kernel_stats.txt | cut -c1-170
rm -rf $O/trace

#!/bin/bash
# round 2, call e: tests of what changed (filtered search, comm exit, relaxed rows), distance-batch variants, PageRank uniform + R-MAT
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_graph.py tests/test_gpu_hnsw.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.txt
timeout 600 python scratch/r2_dist.py > $O/dist.txt 2>&1; echo "dist rc=$?"; cat $O/dist.txt | grep -v amdgpu.ids
timeout 600 python bench.py --skip-hnsw --skip-cpu > $O/pr.json 2> $O/pr.err; echo "pr rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2e/pr.json"))
r = d["pagerank_rmat"]
print("uniform", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "| rmat exact", r["ms_per_iteration"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"],
      "| relaxed", r.get("relaxed", {}).get("ms_per_iteration"), r.get("relaxed", {}).get("roofline", {}).get("avg_launch_ms"), r.get("relaxed", {}).get("error"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o pr -- python $GRAFT_REPO_ROOT/scratch/r2_dist.py > $GRAFT_REPO_ROOT/$O/dist_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/dist_kernel_stats.txt; head -14 $O/dist_kernel_stats.txt | cut -c1-170
rm -rf $O/trace

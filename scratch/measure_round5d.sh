#!/bin/bash
# round 5, final tree, part D: the default workload (no CPU legs) under rocprofv3 --kernel-trace --stats with the search kernel's launches
# listed in launch order (profiles/summarize.py): the timed loop's own average beside the line's avg_launch_ms
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round5d
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "trace rc=$?"
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -8 $O/bench_kernel_stats.txt | cut -c1-170; grep "^#" $O/bench_kernel_stats.txt | tail -40 | cut -c1-200
rm -rf $O/trace
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = json.load(open(R + "/gpurun_out/round5d/bench_under_rocprof.json"))
print("hnsw", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_launch_ms", d["roofline"]["avg_launch_ms"], d["roofline"].get("measured_ceiling"), "wall", d.get("bench_wall_s"))
print("built", d.get("built_handle"))
print("10m clustered", d.get("hnsw_10m_clustered"))
print("box", d["box"].get("pci"), d["box"].get("table_landing"))
PY

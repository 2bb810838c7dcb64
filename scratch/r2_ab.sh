#!/bin/bash
# round 2, call ab: the default bench under rocprofv3 --kernel-trace --stats after the graph-rule work (kernel stats for profiles/)
O=gpurun_out/r2ab; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 1700 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --skip-cpu > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err
echo "trace rc=$?"
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/bench_kernel_stats.txt; head -45 $O/bench_kernel_stats.txt | cut -c1-170
rm -rf $O/trace
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2ab/bench_under_rocprof.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','bench_wall_s')}, d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY

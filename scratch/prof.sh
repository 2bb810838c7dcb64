#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench, then HBM PMC passes (separate runs, as gpurun requires)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench -- python $R/bench.py --skip-cpu > $R/gpurun_out/prof/bench_trace.json 2> $R/gpurun_out/prof/bench_trace.err
echo "trace rc=$?"
ls -R $R/gpurun_out/prof/trace | head -30

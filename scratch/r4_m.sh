#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m
rm -rf $O; mkdir -p $O
cd $R
for warm in copy kernel; do
  CZ_SSSP_WARM=$warm CZ_SSSP_TRACE=1 timeout 1500 python bench.py --skip-cpu > $O/bench_$warm.json 2> $O/bench_$warm.err; echo "$warm rc=$?"
  grep "^sssp mark" $O/bench_$warm.err | sed -n 1,48p | grep -n "warm\|fill dp\|entry" | head -24
done

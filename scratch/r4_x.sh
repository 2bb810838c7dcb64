#!/bin/bash
# configs[4]'s TOTAL graph (100M nodes / 1B edges, what the 8-GPU job shards) on ONE GPU: does the plan build, the sweep and the parity
# with the oracle hold at 10x the bench's size?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4x
rm -rf $O; mkdir -p $O
cd $R
free -g | head -2 | tee $O/mem.txt
timeout 800 python bench.py --skip-hnsw --skip-secondary --pr-nodes 100000000 --pr-edges 1000000000 > $O/bench.json 2> $O/bench.err
echo rc=$?
tail -15 $O/bench.err | grep -v "amdgpu.ids"
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
head -c 3000 $O/bench.json

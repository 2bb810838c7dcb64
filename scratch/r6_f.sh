#!/bin/bash
# round 6: the default bench line at full size, exactly as the driver runs it (mid-round checkpoint)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6f
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -v Warning $O/bench.err | tail -30
cp gpurun_out/bench_detail.json $O/bench_detail.json
echo "line bytes=$(wc -c < $O/bench.json)"

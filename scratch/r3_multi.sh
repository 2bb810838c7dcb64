#!/bin/bash
# the N > 1 code of bench.py with one rank (sharded entry points, collectives, barriers): a smoke run, not a measurement
export TMPDIR=/tmp
mkdir -p gpurun_out/r3multi
CZ_BENCH_FORCE_MULTI=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --n 1000000 --pr-nodes-total 4000000 --pr-edges-total 40000000 --skip-cpu > gpurun_out/r3multi/bench_forced_multi.json 2> gpurun_out/r3multi/bench_forced_multi.err
echo "rc=$?"
grep -v "amdgpu.ids\|Warning" gpurun_out/r3multi/bench_forced_multi.err | tail -8 | cut -c1-250
cut -c1-1500 gpurun_out/r3multi/bench_forced_multi.json

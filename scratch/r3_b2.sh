#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b2
timeout 1200 python bench.py --skip-cpu > gpurun_out/r3b2/bench.json 2> gpurun_out/r3b2/bench.err; echo "rc=$?"
cp gpurun_out/bench_detail.json gpurun_out/r3b2/bench_detail.json
cut -c1-600 gpurun_out/r3b2/bench.json

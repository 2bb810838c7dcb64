#!/bin/bash
# BASELINE.md's own corpus at the metric's N: 10M x 768, 16 clusters (VERDICT r2 missing #6)
export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
timeout 1500 python bench.py --dist clustered --skip-pagerank --skip-secondary --skip-cpu --steps 10 --warmup 2 > gpurun_out/r3t/bench_10m_clustered.json 2> gpurun_out/r3t/bench_10m_clustered.err
echo "rc=$?"
grep -v amdgpu.ids gpurun_out/r3t/bench_10m_clustered.err | tail -12 | cut -c1-300
cut -c1-1500 gpurun_out/r3t/bench_10m_clustered.json

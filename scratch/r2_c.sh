#!/bin/bash
# round 2, call c: gpu tests of what changed, a small bench through every code path, then the default bench (10M x 768)
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_graph.py tests/test_gpu_hnsw_build.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 900 python bench.py --n 2000000 --steps 5 --warmup 2 > $O/bench2m.json 2> $O/bench2m.err; echo "bench2m rc=$?"; tail -30 $O/bench2m.err
timeout 1500 python bench.py > $O/bench10m.json 2> $O/bench10m.err; echo "bench10m rc=$?"; cat $O/bench10m.err | grep -v Warning | tail -40

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1q
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --skip-pagerank --skip-cpu --steps 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac']); print(d['distance_batch'])"
CZ_PAIRS_GROUPED=0 timeout 600 python bench.py --skip-pagerank --skip-cpu --steps 5 > $O/bench0.json 2> $O/bench0.err; python -c "
import json; d=json.load(open('$O/bench0.json')); print('ungrouped', d['distance_batch'])"

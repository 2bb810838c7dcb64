#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round4e
rm -rf $O; mkdir -p $O
cd $R
timeout 120 python bench.py --n 200000 --skip-pagerank --skip-cpu --skip-secondary --steps 5 > $O/small.json 2> $O/small.err; echo "small rc=$?"; grep -v Warning $O/small.err | tail -3
python -c "
import json; d=json.load(open('$O/small.json')); print(d['value'], d['roofline']['frac'], d.get('built_handle'))"
timeout 400 python bench.py --skip-pagerank --skip-cpu --skip-secondary > $O/bench_hnsw.json 2> $O/bench_hnsw.err; echo "10M rc=$?"; grep -v Warning $O/bench_hnsw.err | tail -6
python -c "
import json; d=json.load(open('$O/bench_hnsw.json')); print(d['value'], d['ms_per_step'], d['roofline'], d.get('built_handle'), d.get('distance_batch',{}).get('roofline'), d.get('bench_wall_s'))"

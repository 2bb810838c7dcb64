#!/bin/bash
# round 2, call n: register merge for 512 < ef <= 1024 -- parity tests, then the clustered 1M corpus with and without it
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_hnsw_build.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for v in 1 0; do
  CZ_HNSW_WIDE_MERGE=$v timeout 600 python bench.py --n 1000000 --dist clustered --skip-pagerank --skip-cpu --skip-secondary --ef 768 > $O/clustered_$v.json 2> $O/clustered_$v.err; echo "clustered wide_merge=$v rc=$?"
  python -c "
import json;d=json.load(open('$O/clustered_$v.json'));print('wide_merge=$v', d['ms_per_step'], d['roofline']['frac'], d['config']['recall_at_k'], d['config']['n_dist_per_query'])"
done

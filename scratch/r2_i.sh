#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
for t in test_collectives_of_one_rank "test_pagerank_sharded_and_multi_match_oracle" test_hnsw_search_sharded_one_shard; do
  timeout 300 python -m pytest "tests/test_gpu_comm.py" -k "$t" -m gpu -x -q > $O/$t.txt 2>&1; echo "$t rc=$?"
done
timeout 300 python -m pytest tests/test_gpu_graph.py -m gpu -x -q > $O/graph.txt 2>&1; echo "graph alone rc=$?"
NCCL_DEBUG= timeout 300 python -m pytest tests/test_gpu_comm.py -m gpu -x -q -k "one_shard or collectives" > $O/two.txt 2>&1; echo "two rc=$?"

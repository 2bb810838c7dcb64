#!/bin/bash
# round 2, call x: sharded traversal kernels after the staged frontier build / visited bits (comm tests + one-rank timing)
O=gpurun_out/r2x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_graph.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 600 python scratch/r2_sharded_one_rank.py > $O/one_rank.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/one_rank.txt | tail -10

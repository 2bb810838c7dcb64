#!/bin/bash
# round 2, call ag: write-back of insert / remove as a delta of the stored rows
O=gpurun_out/r2ag; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hnsw_build.py tests/test_zz_stored_index_cpp_gpu.py -m gpu -q -k "write_back or store or remove" > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest.txt

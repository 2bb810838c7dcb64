#!/bin/bash
# round 2, call ah: the C++ mirror's index maintenance + write-back delta on the device
O=gpurun_out/r2ah; mkdir -p $O
timeout 600 python -m pytest tests/test_zz_stored_index_cpp_gpu.py tests/test_cpp_host.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest.txt | cut -c1-200

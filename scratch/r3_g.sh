#!/bin/bash
# round 3, call g: test files touched since call f (C++ host, build/insert/remove with key order), BFS with per-wave chunks of the new-node list
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_cpp_host.py tests/test_gpu_hnsw_build.py tests/test_gpu_graph.py tests/test_mirrors_agree.py tests/test_zz_stored_index_cpp_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 600 python scratch/r3_bfs.py > $O/bfs.txt 2>&1; echo "bfs rc=$?"; grep -v Warning $O/bfs.txt | tail -20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bfs -- python $R/scratch/r3_bfs.py > $R/$O/bfs_traced.txt 2>&1
cd $R
db=$(find $O/trace -name "*.db" | head -1)
python profiles/summarize.py "$db" > $O/kernel_stats.txt; grep -E "bfs_|scan_" $O/kernel_stats.txt | cut -c1-170
rm -rf $O/trace

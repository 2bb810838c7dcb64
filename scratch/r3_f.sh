#!/bin/bash
# round 3, call f: the graph / build test files after the PageRank, BFS, remove and predicate changes; BFS one-pass level A/B; PageRank line
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_hnsw_build.py tests/test_cpp_host.py tests/test_fixed_rule.py tests/test_mirrors_agree.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 600 python scratch/r3_bfs.py > $O/bfs.txt 2>&1; echo "bfs rc=$?"; grep -v Warning $O/bfs.txt | tail -20
timeout 600 python scratch/r3_pr_rmat.py --only-default > $O/pr.txt 2>&1; echo "pr rc=$?"; grep -E "ms/sweep|parity" $O/pr.txt

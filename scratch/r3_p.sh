#!/bin/bash
# round 3, call p: triangles with a 16-lane group per node; BFS with hub lists cut over workgroups (skewed graph)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3p; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q -k "clustering or bfs or shortest" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py 2>&1 | grep -E "clustering|incidences" > $O/tri.txt; cat $O/tri.txt
timeout 600 python scratch/r3_bfs.py > $O/bfs.txt 2>&1; echo "bfs rc=$?"; grep -E "passes=|identical|held|edges, max" $O/bfs.txt

#!/bin/bash
# round 5, call b: the accumulate formulation -- parity tests, then the sweep on the bench graphs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "pagerank" > gpurun_out/r5b/pytest_pr.txt 2>&1
tail -15 gpurun_out/r5b/pytest_pr.txt
timeout 600 python scratch/r5_pr.py both > gpurun_out/r5b/pr.txt 2>&1
cat gpurun_out/r5b/pr.txt

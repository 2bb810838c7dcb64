#!/bin/bash
# round 5: LabelPropagation -- the active set (only dependants of changed nodes once an iteration changed few) and the number of
# nodes a group of the tiny kernel keeps in flight: parity, then the 10M / 200M run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5lp
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "label_prop or degenerate or without_edges" > $O/pytest_lp.txt 2>&1; echo "pytest lp rc=$?"; tail -5 $O/pytest_lp.txt
for u in 1 2 4; do
  CZ_LP_TINY_U=$u timeout 900 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "label_prop" 2>&1 | tail -1
  for f in 0 0.125; do
    echo "== CZ_LP_TINY_U=$u CZ_LP_SPARSE_FRAC=$f"
    CZ_LP_TINY_U=$u CZ_LP_SPARSE_FRAC=$f timeout 600 python scratch/r3_rule_runs.py lp 3 2>&1 | grep -v Warning | tail -3
  done
done | tee $O/lp_runs.txt

#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
for m in load id rank rank_ag multi; do
  timeout 120 python scratch/r2_rccl_exit.py $m > $O/exit_$m.txt 2>&1; echo "mode $m rc=$?"
  CZ_COMM_NO_DESTROY=1 timeout 120 python scratch/r2_rccl_exit.py $m > $O/exit_nd_$m.txt 2>&1; echo "mode $m (no destroy) rc=$?"
  COZO_RCCL_LIB=/opt/rocm/lib/librccl.so.1 COZO_HIP_RUNTIME=system timeout 120 python scratch/r2_rccl_exit.py $m > $O/exit_sys_$m.txt 2>&1; echo "mode $m (system hip+rccl) rc=$?"
done
tail -3 $O/exit_rank.txt $O/exit_sys_rank.txt
timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest graph rc=$?"; tail -5 $O/pytest.txt
timeout 600 python scratch/graph_rules_bench.py > $O/graph_rules.txt 2>&1; echo "graph rules rc=$?"; cat $O/graph_rules.txt | grep -v amdgpu.ids

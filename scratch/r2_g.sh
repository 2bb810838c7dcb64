#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
for m in torch_only rank_torch rank_ag_stream rank_ar sharded sharded_torch rank_ag; do
  timeout 120 python scratch/r2_rccl_exit.py $m > $O/exit_$m.txt 2>&1; echo "mode $m rc=$?"
done
tail -4 $O/exit_rank_ag.txt

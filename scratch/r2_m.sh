#!/bin/bash
# round 2, call m: what one big SSSP relaxation round spends its time on
O=gpurun_out/r2m; mkdir -p $O
CZ_SSSP_EXPERIMENT=1 timeout 600 python scratch/sssp_experiment.py > $O/experiment.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/experiment.txt | tail -60

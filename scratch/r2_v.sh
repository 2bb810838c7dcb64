#!/bin/bash
# round 2, call v: configs[3] (10M in 8 sub-indices + merge) emulated on one GPU
O=gpurun_out/r2v; mkdir -p $O
timeout 1200 python scratch/r2_cfg3_emulated.py > $O/cfg3.txt 2> $O/cfg3.err
echo "rc=$?"; tail -3 $O/cfg3.err; grep -v amdgpu.ids $O/cfg3.txt | cut -c1-400 | tail -16

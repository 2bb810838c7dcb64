#!/bin/bash
# round 2, call s: pair order vs cz_distance_batch time (TLB / DRAM locality at 30 GB?)
O=gpurun_out/r2s; mkdir -p $O
timeout 600 python scratch/r2_dist_order.py > $O/dist_order.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/dist_order.txt | tail -12

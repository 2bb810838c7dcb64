#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1r
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python scratch/pr_shard.py > $O/pr_shard.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pr_shard.txt | tail -20

#!/bin/bash
# round-2 probe: the metric's own configuration (10M x 768) on the round-1 kernels
O=gpurun_out/r2a; mkdir -p $O
nproc > $O/host.txt; free -g >> $O/host.txt; rocm-smi --showmeminfo vram >> $O/host.txt 2>&1
timeout 1500 python bench.py --n 10000000 --skip-pagerank --steps 10 --warmup 2 > $O/bench10m.json 2> $O/bench10m.err
echo "rc=$?"; tail -20 $O/bench10m.err; cat $O/bench10m.json

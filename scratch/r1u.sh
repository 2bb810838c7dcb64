#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1u
rm -rf $O; mkdir -p $O
cd $R
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power \(W\)|socclk" ; sleep 0.5; done ) > $O/smi.txt 2>&1 &
SMI=$!
for i in 1 2 3 4; do
  echo "run $i start $(date +%s.%N)" >> $O/marks.txt
  timeout 600 python bench.py --skip-pagerank --skip-cpu --steps 40 --ef 96 > $O/b$i.json 2> $O/b$i.err
  echo "run $i end $(date +%s.%N)" >> $O/marks.txt
  python -c "
import json; d=json.load(open('$O/b$i.json')); print('run $i', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), 'dist', round(d['distance_batch']['ms'],3), 'build', round(d['config']['index_build_s'],1))"
done
kill $SMI
cat $O/marks.txt
grep -c sclk $O/smi.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4y
rm -rf $O; mkdir -p $O
cd $R
for v in "CZ_PR_FLAT=1" "CZ_PR_FLAT=0" "CZ_PR_MODE=gather"; do
  echo "== $v" | tee -a $O/out.txt
  env $v timeout 400 python bench.py --skip-hnsw --skip-secondary --skip-cpu --pr-nodes 100000000 --pr-edges 1000000000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])" | tee -a $O/out.txt
done

#!/bin/bash
# round 6, part H: BFS hub levels (aggregated tally / place, workgroup order_big, batched descriptors in the long kernel): BFS tests,
# R-MAT and uniform rule legs, trace; the long-list threshold 4096 (default build) / 1024 / 512 (scratch/lib variants)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round6h
rm -rf $O; mkdir -p $O
cd $R
for lib in ""; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  echo "## lib=${lib:-default}"
  COZO_GPU_LIB=${lib:+$R/$lib} timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_sharded_driver.py -q -m gpu -k "bfs or resident or rules_on or shortest" 2>&1 | tail -1
  for i in 1 2; do COZO_GPU_LIB=${lib:+$R/$lib} timeout 300 python scratch/r6_rules.py rmat 2>&1 | grep -E "^bfs " | cut -c1-100; done
  COZO_GPU_LIB=${lib:+$R/$lib} timeout 300 python scratch/r6_rules.py uniform 2>&1 | grep -E "^bfs " | cut -c1-100
done 2>&1 | tee $O/bfs_variants.txt
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/scratch/r6_rules.py rmat > $O/rules_rmat.txt 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/profiles/summarize.py "$db" > $O/rmat_kernel_stats.txt
grep -E "^bfs" $O/rmat_kernel_stats.txt | cut -c1-150
grep -E "^bfs " $O/rules_rmat.txt | cut -c1-120
rm -rf $O/trace

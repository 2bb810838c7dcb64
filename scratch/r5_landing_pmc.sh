#!/bin/bash
# for the round after 4: what differs between a good and a bad landing of the 30 GB vector table (profiles/r04_built_vs_created.txt)?
# Four tables held at once (vmm_bench hold:4: per-table rates repeat to 0.1 % and differ by 1.5-4 %), distance_pairs_kernel over each in
# turn, once per counter set; the per-dispatch counters line up with the per-table times by dispatch order.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_landing
rm -rf $O; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scratch/vmm_bench.hip -Iinclude -Lcozo_amd/lib -lcozo_gpu -Wl,-rpath,'$ORIGIN/../cozo_amd/lib' -o scratch/vmm_bench
scratch/vmm_bench 10000000 6 hold:4 | tee $O/plain.txt
cd /tmp && export TMPDIR=/tmp
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "distance_pairs_kernel" --output-format csv -d $O/pmc_$tag -o pmc -- $R/scratch/vmm_bench 10000000 2 hold:4 > $O/pmc_$tag.out 2>&1
  echo "pmc $tag rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for f in sorted(glob.glob(R + "/gpurun_out/r5_landing/pmc_*/**/*counter_collection.csv", recursive=True)):
    rows = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        rows[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in rows.items():
        # dispatches: per round (3) x per table (4) x (2 warm + 2 timed)
        per = [sum(v[i:i + 4]) / 4 for i in range(0, len(v), 4)]
        print(k, " ".join(f"{x:.4g}" for x in per))
PY

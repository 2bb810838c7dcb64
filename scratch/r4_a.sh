#!/bin/bash
# round 4, call A: what does placement of the 30 GB table do to distance_pairs_kernel?  + what the box reports about itself
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a
rm -rf $O; mkdir -p $O
cd $R
{
echo "== rocm-smi"; rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmeminfo vram --showperflevel 2>&1 | grep -v "^$" | head -60
echo "== sysfs"
for c in /sys/class/drm/card*/device; do
  [ -e $c/current_memory_partition ] || [ -e $c/pp_dpm_sclk ] || continue
  echo "-- $c"
  for f in current_memory_partition current_compute_partition available_memory_partition mem_info_vram_total mem_info_vram_used pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level; do
    [ -r $c/$f ] && { echo "[$f]"; cat $c/$f 2>&1 | head -12; }
  done
  for h in $c/hwmon/hwmon*; do for f in power1_average power1_input power1_cap freq1_input freq2_input temp1_input temp2_input temp3_input; do [ -r $h/$f ] && echo "$f=$(cat $h/$f 2>&1)"; done; done
done
echo "== kernel params"; cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/vm_size 2>&1
uname -r; nproc
} > $O/box.txt 2>&1
timeout 300 scratch/vmm_bench 10000000 10 malloc vmm:2:2 vmm:1024:1024 vmm1:1024 vmm1:2 malloc window:1000000 malloc vmm:1024:1024 > $O/vmm_fresh.txt 2>&1; echo "fresh rc=$?"
cat $O/vmm_fresh.txt
timeout 400 scratch/vmm_bench 10000000 10 frag:200:8 malloc vmm:1024:1024 vmm:2:2 unfrag malloc > $O/vmm_frag.txt 2>&1; echo "frag rc=$?"
cat $O/vmm_frag.txt
cd /tmp && export TMPDIR=/tmp
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"; do
  tag=$(echo $set | cut -c1-40 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "distance_pairs_kernel" --output-format csv -d $O/pmc_$tag -o pmc -- $R/scratch/vmm_bench 10000000 2 malloc vmm:1024:1024 window:1000000 malloc > $O/pmc_$tag.out 2>&1
  echo "pmc $tag rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4a"
for f in sorted(glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for k, v in acc.items():
        v.sort()
        print(k, " ".join(f"{x[1]:.4g}" for x in v))
PY

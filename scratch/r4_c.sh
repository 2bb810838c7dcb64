#!/bin/bash
# round 4, call C: the whole bench under a UTCL1 counter pass -- translation misses of hnsw_knn_kernel / distance_pairs_kernel at 10M and of
# sssp_relax_kernel in the cold call and in the held call that ran 5x slower inside the whole bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-include-regex "sssp_relax_kernel|hnsw_knn_kernel|distance_pairs_kernel" --output-format csv -d $O/pmc -o pmc -- python $R/bench.py --skip-cpu > $O/bench_under_pmc.json 2> $O/bench_under_pmc.err
echo "rc=$?"
cp $R/gpurun_out/bench_detail.json $O/bench_detail.json
grep -v Warning $O/bench_under_pmc.err | tail -15
python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4c"
rows = collections.defaultdict(dict)
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-60:]
        rows[(int(r["Dispatch_Id"]), k)][r["Counter_Name"]] = float(r["Counter_Value"])
with open(O + "/utcl1_per_dispatch.txt", "w") as out:
    for (d, k), c in sorted(rows.items()):
        req, miss = c.get("TCP_UTCL1_REQUEST_sum", 0), c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0)
        out.write(f"{d:7d} {k:60s} req {req:.4g} miss {miss:.4g} rate {miss / req if req else 0:.5f}\n")
# summary: per kernel, groups of consecutive dispatches
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (d, k), c in rows.items():
    a = agg[k]; a[0] += 1; a[1] += c.get("TCP_UTCL1_REQUEST_sum", 0); a[2] += c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0)
for k, a in agg.items():
    print(k, "dispatches", a[0], "req", f"{a[1]:.4g}", "miss", f"{a[2]:.4g}", "rate", f"{a[2] / a[1] if a[1] else 0:.5f}")
d = json.load(open(O + "/bench_detail.json"))
print("sssp", json.dumps(d.get("graph_rules", {}).get("sssp", {}))[:600])
print("box", json.dumps(d.get("box", {}))[:900])
PY
rm -rf $O/pmc

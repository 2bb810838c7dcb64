"""gpurun_out/round/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE}/**/*counter_collection.csv (rocprofv3 --pmc, separate passes, no tracing:
scratch/measure_round.sh) -> profiles/pmc_traffic.json, the per-launch HBM bytes bench.py reports as `roofline.traffic`.

gfx950 corrections (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on pb_expand_kernel, whose known reads are reported
at exactly half): FETCH_SIZE x 2, WRITE_SIZE as reported; both are KiB per dispatch.  Every entry carries the hash of the device
sources it was measured on (bench.PMC_SOURCES): bench.py reports null when they have changed since.

    python profiles/make_pmc_traffic.py [round_dir]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# key -> (pmc tag, kernel regex, how the dispatches of the tagged run map onto ONE launch of the workload, algorithmic bytes)
#   "last": the average of the last `n` dispatches of each kernel (the timed-loop launches), summed over the kernels
#   "per_run": every dispatch matching the regex, summed, divided by `runs` (a rule = many kernels per call)
SPEC = {
    "hnsw_knn": dict(tag="hnsw", regex=r"hnsw_knn_kernel", mode="last", n=3, algo=None),
    "distance_batch": dict(tag="hnsw", regex=r"distance_pairs_kernel", mode="last", n=3, algo=(1 << 22) * 4 * 768),
    # (the plan chooses the formulation: an entry is written only when its phase-B kernel ran in the tagged pass)
    "pagerank_blocked": dict(tag="pr", regex=r"pb_expand_kernel|pb_reduce_kernel", mode="last", n=3, algo=None, require=r"pb_reduce_kernel", forbid=r"pa_reduce_kernel"),
    "pagerank_accumulate": dict(tag="pr", regex=r"pb_expand_kernel|pa_reduce_kernel|pb_reduce_kernel", mode="last", n=3, algo=None, require=r"pa_reduce_kernel"),
    "pagerank_blocked_rmat": dict(tag="prrmat", regex=r"pb_expand_kernel|pb_reduce_kernel|pr_hub_kernel|pr_empty_rows_kernel", mode="last", n=3, algo=None),
    # the in-place reading: init (one launch of every phase-A item) + 5 sweeps of L + 2 launches (scratch/r6_inplace.py, IP_FEW=1)
    "pagerank_inplace": dict(tag="prip", regex=r"gi_level_kernel|gi_long_kernel|gi_sum_partials_kernel", mode="per_run", runs=5, algo=None,
                             skip_first={"gi_level_kernel": 1}),
    "hnsw_knn_1m": dict(tag="hnsw1m", regex=r"hnsw_knn_kernel", mode="last", n=3, algo=None),
    "bfs": dict(tag="bfs", regex=r"bfs_|scan_tiles_kernel|scan_add_kernel", mode="per_run", runs=2, algo=None),
    "sssp": dict(tag="sssp", regex=r"sssp_|fill_u64_kernel", mode="per_run", runs=2, algo=None),
    "connected_components": dict(tag="cc", regex=r"cc_|scan_tiles_kernel|scan_add_kernel", mode="per_run", runs=2, algo=None),
    "clustering_coefficients": dict(tag="tri", regex=r"triangles_|tri_", mode="per_run", runs=2, algo=None),
    "label_propagation": dict(tag="lp", regex=r"lp_|iota_kernel|scan_tiles_kernel|scan_add_kernel", mode="per_run", runs=2, algo=None),
}


RANDOM_ACCESS = {"bfs", "sssp", "connected_components", "clustering_coefficients", "label_propagation"}


def load(round_dir, tag, counter):
    per_kernel = collections.defaultdict(list)
    for f in glob.glob(os.path.join(round_dir, f"pmc_{tag}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
            per_kernel[k].append(float(row["Counter_Value"]))
    return per_kernel


def main():
    round_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "round")
    algos = {}
    try:  # algorithmic bytes of the workloads, from the bench line of the same round
        d = json.load(open(os.path.join(round_dir, "bench_detail.json")))
        algos["hnsw_knn"] = d["roofline"]["algorithmic_bytes_per_launch"]
        algos["pagerank_blocked"] = algos["pagerank_accumulate"] = d["pagerank"]["roofline"]["algorithmic_bytes_per_launch"]
        algos["pagerank_inplace"] = d["pagerank"]["roofline"]["algorithmic_bytes_per_launch"]
        algos["pagerank_blocked_rmat"] = d["pagerank_rmat"]["roofline"]["algorithmic_bytes_per_launch"]
        algos["hnsw_knn_1m"] = d["hnsw_1m"]["roofline"]["algorithmic_bytes_per_launch"]
        algos["bfs"] = d["graph_rules"]["bfs"]["algorithmic_bytes"]
        algos["sssp"] = d["graph_rules"]["sssp"]["algorithmic_bytes"]
        for k in ("connected_components", "clustering_coefficients", "label_propagation"):
            algos[k] = d["graph_rules"][k]["algorithmic_bytes"]
    except Exception as e:  # noqa: BLE001
        print("no bench_detail.json beside the PMC passes:", e)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate runs, no tracing (scratch/measure_round.sh; this file is written by "
                   "profiles/make_pmc_traffic.py from the per-dispatch CSVs).  FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at "
                   "64 B, MI355X_MICROARCH.md HBM section; calibrated in round 1 on pb_expand_kernel, whose known reads 200 MB ids + 160 MB "
                   "slice staging = 360 MB are reported as 178 MiB), WRITE_SIZE as reported (392 MiB vs 400 MB known).  Every entry carries "
                   "the hash of the device sources it was measured on (bench.py PMC_SOURCES); bench.py reports traffic = null when they "
                   "have changed since, or when the launch's algorithmic bytes differ from the profiled launch's by more than 2 %."}
    try:  # entries whose passes are not under round_dir this time stay as they are (bench.py checks their source hash)
        with open(path) as f:
            for k, v in json.load(f).items():
                if k != "_how":
                    out[k] = v
    except Exception:  # noqa: BLE001
        pass
    for key, sp in SPEC.items():
        parts, total, ok = {}, 0.0, True
        for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            pk = load(round_dir, sp["tag"], counter)
            rx = re.compile(sp["regex"])
            hit = {k: v for k, v in pk.items() if rx.search(k)}
            if not hit or (sp.get("require") and not any(re.search(sp["require"], k) for k in pk)) or \
                    (sp.get("forbid") and any(re.search(sp["forbid"], k) for k in pk)):
                ok = False
                break
            for k, vals in sorted(hit.items()):
                vals = vals[sp.get("skip_first", {}).get(k, 0):]
                kib = sum(vals[-sp["n"]:]) / len(vals[-sp["n"]:]) if sp["mode"] == "last" else sum(vals) / sp["runs"]
                parts[f"{k} {counter}"] = kib
                total += kib * 1024.0 * scale
        if not ok:
            print(f"{key}: no PMC rows under pmc_{sp['tag']}_*: entry left as it was")
            continue
        algo = sp["algo"] or algos.get(key)
        out[key] = dict(bytes_per_launch=int(total), parts_KiB={k: round(v, 1) for k, v in parts.items()}, algorithmic_bytes=algo,
                        source_hash=bench.kernel_source_hash(key))
        if key in RANDOM_ACCESS:  # the x2 is calibrated on wide coalesced streams only
            out[key]["calibration"] = ("UPPER BOUND: FETCH_SIZE x 2 is calibrated on 128-byte coalesced requests; these kernels issue 4- / 8-byte "
                                       "random accesses, for which the guide gives no factor -- good to about 2x, not a measurement")
        print(f"{key}: {total / 1e9:.3f} GB per launch" + (f" = {total / algo:.3f} x the algorithmic bytes" if algo else ""))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()

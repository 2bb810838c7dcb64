"""bench_detail.json (the unabridged objects bench.py writes beside its line) -> the measurement table of DESIGN.md section 5.

    python profiles/make_measurements.py profiles/r06_bench_detail.json [--graph-legs profiles/r06_bench_detail_graph_legs.json] [--write]

Prints the markdown table; with --write it replaces the text between the GENERATED markers in DESIGN.md (the table is never edited
by hand: VERDICT r5 item 2)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def g(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def f(x, nd=3, unit=""):
    if x is None:
        return "–"
    if isinstance(x, bool):
        return "yes" if x else "NO"
    if isinstance(x, (int, float)):
        ax = abs(x)
        if ax >= 1e9:
            return f"{x / 1e9:.{nd}g} G{unit}"
        if ax >= 1e6:
            return f"{x / 1e6:.{nd}g} M{unit}"
        if ax >= 1e3:
            return f"{x / 1e3:.{nd}g} k{unit}"
        return f"{x:.{nd}g}{unit}"
    return str(x)


def traffic(rf, algo=None):
    t = (rf or {}).get("traffic")
    a = (rf or {}).get("algorithmic_bytes_per_launch") or algo
    if not t:
        return "–"
    return f"{t / 1e9:.2f} GB" + (f" = {t / a:.2f} ×" if a else "")


def rows(d):
    out = []
    rf = d.get("roofline", {})
    cb = d.get("cpu_baseline", {})
    par = d.get("parity", {})
    out.append(("**" + str(g(d, "config", "workload", default="HNSW k-NN 10M x 768")) + "** (`hnsw_knn_kernel`), ef " + f(g(d, "config", "ef")) + ", recall@10 " + f(g(d, "config", "recall_at_k"), 4),
                f(d.get("value"), 4, " queries/s") + f", {f(d.get('ms_per_step'), 4)} ms", f(rf.get("frac"), 3), traffic(rf),
                f(cb.get("value"), 3, " q/s") + f" on {cb.get('cores')} core(s)" + (f"; {f(g(cb, 'all_cores', 'value'), 3)} on {g(cb, 'all_cores', 'cores')}" if cb.get("all_cores") else ""),
                "bit-equal to the oracle: " + f(par.get("bit_equal_to_oracle", par.get("parity_checked"))) + f"; max rel err vs reference order {f(par.get('max_rel_err_vs_reference_arithmetic'))}"))
    db = d.get("distance_batch") or {}
    out.append(("batched distance, 4M pairs on the index's settled table (`distance_pairs_kernel`)", f(db.get("distances_per_s"), 4, " dist/s") + f", {f(db.get('ms'), 4)} ms",
                f(g(db, "roofline", "frac"), 3) + f" (bare table {f(g(db, 'bare_table', 'frac'), 3)})", traffic(db.get("roofline")), "–",
                "same bits as the bare-table call: " + f(db.get("same_bits_as_bare_table"))))
    ex = d.get("exact_scan") or {}
    out.append(("exhaustive scan as an f32 MFMA GEMM (`dot_gemm_mfma_kernel`)", f(ex.get("queries_per_s"), 4, " queries/s") + f", {f(ex.get('ms_per_batch'), 4)} ms",
                f(g(ex, "roofline", "frac"), 3) + " of 157.3 TFLOP/s", "–", "–", "recall 1.0 (the ground truth)"))
    for key, label in (("pagerank", "uniform"), ("pagerank_rmat", "R-MAT")):
        pr = d.get(key) or {}
        if not pr:
            continue
        out.append((f"**PageRank {f(pr.get('nodes'), 3)} / {f(pr.get('edges'), 3)} {label}, Jacobi reading** ({pr.get('form')})", f(pr.get("value"), 4, " edges/s") + f", {f(g(pr, 'roofline', 'avg_launch_ms'), 4)} ms/sweep (events)",
                    f(g(pr, "roofline", "frac"), 3), traffic(pr.get("roofline")),
                    f(g(pr, "cpu_baseline", "value"), 3, " edges/s") + f" on {g(pr, 'cpu_baseline', 'cores')} threads",
                    "3 sweeps == oracle: " + f(g(pr, "parity", "parity_checked"))))
        ip = pr.get("inplace_reading") or {}
        if "roofline" in ip:
            out.append((f"**PageRank {f(pr.get('nodes'), 3)} / {f(pr.get('edges'), 3)} {label}, in-place reading** (`gi_level_kernel`, {g(ip, 'plan', 'launches_per_sweep')} launches / sweep)",
                        f(ip.get("value"), 4, " edges/s") + f", {f(g(ip, 'roofline', 'avg_launch_ms'), 4)} ms/sweep (events; spread {f(ip.get('event_spread'), 2)})",
                        f(g(ip, "roofline", "frac"), 3), traffic(ip.get("roofline")),
                        f(g(ip, "cpu_baseline", "value"), 3, " edges/s") + " on 1 thread", "3 sweeps == oracle (in-place mode): " + f(g(ip, "parity", "parity_checked"))))
    for leg, label in (("graph_rules", "uniform"), ("graph_rules_rmat", "R-MAT")):
        gr = d.get(leg) or {}
        for k, name in (("bfs", "BFS"), ("connected_components", "ConnectedComponents"), ("sssp", "ShortestPathDijkstra"),
                        ("clustering_coefficients", "ClusteringCoefficients"), ("label_propagation", "LabelPropagation")):
            o = gr.get(k) or {}
            if not o:
                continue
            cbl = o.get("cpu_baseline") or {}
            out.append((f"{name}, {label}", f"{f(o.get('device_ms'), 4)} ms device ({f(o.get('wall_ms'), 4)} ms call)" if "device_ms" in o else str(o.get("cancelled") or o.get("error")),
                        f(g(o, "roofline", "frac"), 2) + (f"; {f(o.get('random_frac'), 2)} of the random-access rate" if o.get("random_frac") else ""),
                        traffic(o.get("roofline"), o.get("algorithmic_bytes")),
                        (f(cbl.get("value"), 3, " edges/s") + " on 1 thread") if cbl.get("value") else str(cbl.get("skipped") or cbl.get("error") or "–")[:60],
                        f(o.get("parity_checked")) if "parity_checked" in o else str((o.get("parity") or {}).get("skipped", "–"))[:60]))
    for key, label in (("hnsw_1m", "HNSW 1M × 768 (configs[1])"), ("hnsw_1m_clustered", "HNSW 1M, 16-cluster corpus"), ("hnsw_10m_clustered", "HNSW 10M, 16-cluster corpus")):
        o = d.get(key) or {}
        if not o:
            continue
        out.append((label + (f", ef {o.get('ef')}, recall {f(o.get('recall_at_k', o.get('recall')), 4)}" if "ef" in o else ""),
                    f(o.get("value"), 4, " queries/s") if "value" in o else str(o.get("skipped") or o.get("error"))[:80],
                    f(g(o, "roofline", "frac"), 3), traffic(o.get("roofline")), "–",
                    ("exact scan: " + f(g(o, "exact_scan", "queries_per_s"), 4, " q/s")) if o.get("exact_scan") else "–"))
    hi = d.get("host_ingest") or {}
    if hi:
        out.append(("host ingest (stored rows → ids + CSR)", f(hi.get("rows_per_s", hi.get("value")), 4, " rows/s"), "–", "–", "–", "–"))
    return out


def table(d, src, note=""):
    lines = [f"_Generated by `profiles/make_measurements.py` from `{src}` (box {g(d, 'box', 'pci', default='?')}, bench wall {f(d.get('bench_wall_s'), 4)} s){note}. "
             "Roofline = algorithmic bytes ÷ event-timed launch ÷ 8 TB/s (MFMA row: ÷ 157.3 TFLOP/s); PMC traffic = FETCH_SIZE × 2 + WRITE_SIZE per launch "
             "(`profiles/pmc_traffic.json`); CPU = the oracle (a C port of the reference's loop) on the same box._", "",
             "| line | value | roofline frac | PMC traffic | CPU oracle | parity |", "|---|---|---|---|---|---|"]
    for r in rows(d):
        lines.append("| " + " | ".join(str(c).replace("|", "/") for c in r) + " |")
    return "\n".join(lines)


def main():
    path = sys.argv[1]
    d = json.load(open(path))
    note = ""
    if "--graph-legs" in sys.argv:  # the graph-rule legs of a later `bench.py --skip-hnsw` run (graph.hip changed after the full run)
        gp = sys.argv[sys.argv.index("--graph-legs") + 1]
        gd = json.load(open(gp))
        for leg in ("graph_rules", "graph_rules_rmat"):
            d[leg] = gd[leg]
        note = (f"; the BFS … LabelPropagation rows from `{os.path.relpath(os.path.abspath(gp), ROOT)}`, a `bench.py --skip-hnsw` run of the "
                f"final `graph.hip`" + (f" (box {g(gd, 'box', 'pci')})" if g(gd, 'box', 'pci', default=None) else ""))
    t = table(d, os.path.relpath(os.path.abspath(path), ROOT), note)
    print(t)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, b = "<!-- BEGIN GENERATED MEASUREMENTS -->", "<!-- END GENERATED MEASUREMENTS -->"
        i, j = s.index(a) + len(a), s.index(b)
        open(p, "w").write(s[:i] + "\n" + t + "\n" + s[j:])


if __name__ == "__main__":
    main()

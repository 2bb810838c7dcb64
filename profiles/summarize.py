#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / avg / min / max.

    python profiles/summarize.py gpurun_out/prof/trace/bench_results.db > profiles/rNN_<what>_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats  (source: {path})")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; durations in microseconds")
    print("%-72s %7s %12s %11s %11s %11s %6s %5s %5s %6s %7s" %
          ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds_B"))
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, scr in rows[:40]:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short[:70]
        print("%-72s %7d %12.1f %11.2f %11.2f %11.2f %6.2f %5s %5s %5s %7s" %
              (short, n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, lds))
    # the two kernels the bench's roofline objects are about, by launch shape
    for pat in ("%hnsw_knn_kernel%", "%hnsw_knn_spec_kernel%", "%hnsw_knn_wide_kernel%", "%gi_level_kernel%", "%pr_step_kernel%", "%pb_expand_kernel%", "%pb_reduce_kernel%", "%pa_reduce_kernel%", "%distance_pairs_kernel%", "%distance_runs_kernel%", "%dot_gemm_mfma_kernel%", "%bf_select_kernel%"):
        for g, lds, n, avg, mn, mx in c.execute(
                "select grid_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels "
                "where name like ? group by grid_x, lds_size order by count(*) desc", (pat,)):
            print(f"# {pat.strip('%')}: grid_x={g} lds={lds} launches={n} avg={avg / 1e3:.2f}us min={mn / 1e3:.2f}us max={mx / 1e3:.2f}us")
    # The search kernel in LAUNCH ORDER, as runs of consecutive launches of one shape and (within 12 %) one duration: the timed loop of
    # the bench is the run of warm-up + steps launches (5 + 20 by default) -- its average is what `roofline.avg_launch_ms` has to agree with.
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    order = "start" if "start" in cols else "rowid"
    runs, cur = [], None
    for g, lds, dur in c.execute(f"select grid_x, lds_size, duration from kernels where name like '%hnsw_knn_kernel%' order by {order}"):
        if cur and cur["g"] == g and cur["lds"] == lds and abs(dur - cur["d"][0]) <= 0.12 * cur["d"][0]:
            cur["d"].append(dur)
        else:
            cur = dict(g=g, lds=lds, d=[dur])
            runs.append(cur)
    print("# hnsw_knn_kernel in launch order: runs of >= 10 consecutive launches of one shape and duration")
    for r in runs:
        if len(r["d"]) >= 10:
            d = r["d"]
            print(f"#   {len(d):4d} launches  grid_x={r['g']} lds={r['lds']}  avg={sum(d) / len(d) / 1e3:.2f}us  min={min(d) / 1e3:.2f}us  max={max(d) / 1e3:.2f}us")


if __name__ == "__main__":
    main(sys.argv[1])

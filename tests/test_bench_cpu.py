"""bench.py pieces that do not need a device: argument defaults (the driver's contract: --gpus / --steps / --warmup, N = 1 by
default) and the host_ingest leg."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_argument_contract(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.n == 10_000_000 and a.dim == 768 and a.batch == 1024 and a.k == 10
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)


def test_host_ingest_leg_runs_without_a_device():
    out = _bench().bench_host_ingest(n_rows=30_000, n_nodes=3_000)
    json.dumps(out)  # goes into the bench line
    assert out["rows"] > 20_000 and out["nodes"] == 3_000 and out["rows_per_s"] > 0
    assert out["id_assignment_s"] > 0 and out["csr_both_s"] > 0 and out["threads"] >= 1


def test_pmc_traffic_is_tied_to_kernels_and_workload():
    """roofline.traffic comes from the committed PMC summary only for the kernels AND the workload it was measured on"""
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        d = json.load(f)
    for key in ("hnsw_knn", "distance_batch", "pagerank_blocked"):
        ent = d[key]
        if ent["source_hash"] != b.kernel_source_hash(key):  # the kernels changed since the summary was taken: no traffic figure
            assert b.pmc_traffic(key, 1, ent["algorithmic_bytes"]) is None
            continue
        assert b.pmc_traffic(key, 1, ent["algorithmic_bytes"]) == ent["bytes_per_launch"]
        assert b.pmc_traffic(key, 1, ent["algorithmic_bytes"] * 10) is None   # another size: not this measurement
        assert b.pmc_traffic(key, 8, ent["algorithmic_bytes"]) is None        # N > 1: not this measurement
        assert ent["bytes_per_launch"] >= ent["algorithmic_bytes"] * 0.98     # traffic below the algorithmic bytes would be a model error


def test_thread_ladder_is_sane():
    ladder, usable = _bench().thread_ladder()
    assert usable >= 1 and ladder == sorted(set(ladder)) and ladder[-1] == usable and all(1 <= t <= usable for t in ladder)


def test_committed_line_fits_the_drivers_tail_and_keeps_the_contract():
    """the line of the last full run on record (profiles/r04_bench_final.json) re-printed by today's bench_line: under the limit,
    `roofline` and `cpu_baseline` whole, and a `built_handle` object (the build's own handle, timed beside the re-created index)
    survives the cut with its numbers"""
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r04_bench_final_detail.json")) as f:
        full = json.load(f)
    full["built_handle"] = dict(ms_per_step=5.44, frac=0.634, avg_launch_ms=5.43, same_results_after_reload=True, reload_s=1.2,
                                what="x" * 400)
    txt, _ = b.bench_line(full)
    line = json.loads(txt)
    assert len(txt) <= b.LINE_LIMIT
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["built_handle"]["frac"] == 0.634 and line["built_handle"]["same_results_after_reload"] is True
    assert "what" not in line["built_handle"]  # prose stays in the detail file
    assert "model" not in line["config"] and "workload" in line["config"]


def test_reload_flags_default_to_the_boundary_path(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.no_reload is False and a.index_cache is None
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-reload", "--index-cache", "/tmp/x"])
    a = b.parse()
    assert a.no_reload is True and a.index_cache == "/tmp/x"

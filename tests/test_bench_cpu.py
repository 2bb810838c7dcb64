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

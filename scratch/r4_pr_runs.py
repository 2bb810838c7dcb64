"""scratch (round 4): phase B of the blocked PageRank sweep against the LENGTH of its (row block, slice) runs.  The run length is
tile x slice / N = 16384 x 2^wlog / 10M values; CZ_PR_SLICE_LOG2 = 15 (shipped: 53.7 values = 215 B) / 14 / 13 / 12 halves it each time.
Times by HIP events here; the same process under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE gives the bytes per sweep (5 sweeps per setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scratch"))
from cozo_amd import _lib
L = _lib.lib()
import torch
from cozo_amd.graph import PageRankPlan
from pr_sweep import gen

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
n, e = 10_000_000, 100_000_000
off, s, od, E = gen(n, e, dev)
stream = torch.cuda.current_stream().cuda_stream
ref = None
for wlog in (15, 14, 13, 12):
    os.environ["CZ_PR_SLICE_LOG2"] = str(wlog)
    plan = PageRankPlan(off, s, od, n, 0, n, 0.85, device_ptrs=True, mode="blocked")
    c0 = torch.empty(n, dtype=torch.float32, device=dev); c1 = torch.empty_like(c0)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    plan.init(c0, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan.step(c0, c1, err, stream); c0, c1 = c1, c0
    e0.record()
    for _ in range(4):
        plan.step(c0, c1, err, stream); c0, c1 = c1, c0
    e1.record(); torch.cuda.synchronize()
    sc = plan.read_scores()
    same = None if ref is None else bool((sc == ref).all())
    if ref is None: ref = sc
    print(f"slice 2^{wlog}: run = {16384 * (1 << wlog) / n:6.1f} values = {16384 * (1 << wlog) / n * 4:6.0f} B   {e0.elapsed_time(e1) / 4:.4f} ms per sweep   scores_same={same}", flush=True)
    plan.close()

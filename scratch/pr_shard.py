"""scratch: one rank's sweep of a row-sharded PageRank (weak scaling): rows [0, R) of an N-node graph, E local in-edges whose
sources span all N nodes.  Compares the formulations at the shapes bench.py --gpus 2/4/8 gives one rank."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
from cozo_amd.graph import PageRankPlan

def gen(n_total, rows, e, device):
    g = torch.Generator(device=device); g.manual_seed(4242)
    dst = torch.randint(0, rows, (e,), generator=g, device=device, dtype=torch.int64)
    src = torch.randint(0, n_total, (e,), generator=g, device=device, dtype=torch.int64)
    key = torch.unique(dst * n_total + src)
    d = torch.div(key, n_total, rounding_mode="floor")
    s = (key - d * n_total).to(torch.int32)
    off = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(torch.bincount(d, minlength=rows), 0)
    outdeg = torch.bincount(s.to(torch.int64), minlength=n_total).clamp_(min=1).to(torch.int32)
    return off.to(torch.int32), s, outdeg, int(off[-1].item())

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    rows, e = 10_000_000, 100_000_000
    stream = torch.cuda.current_stream().cuda_stream
    for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
        n = rows * world
        off, s, od, E = gen(n, rows, e, dev)
        ref = None
        for mode, env in [("auto", {}), ("blocked", {"CZ_PR_FLAT": "0"}), ("blocked", {"CZ_PR_FLAT": "1"}), ("gather", {})]:
            if world == 1 and mode == "gather": continue
            for k in ("CZ_PR_XCD", "CZ_PR_FLAT"): os.environ.pop(k, None)
            os.environ.update(env)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            plan = PageRankPlan(off, s, od, n, 0, rows, 0.85, device_ptrs=True, mode=None if mode == "auto" else mode)
            torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
            c0 = torch.empty(n, dtype=torch.float32, device=dev); c1 = torch.empty_like(c0)
            err = torch.zeros(1, dtype=torch.float64, device=dev)
            plan.init(c0, stream)
            for _ in range(2):
                plan.step(c0, c1, err, stream); c1[rows:] = c0[rows:]; c0, c1 = c1, c0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 6
            e0.record()
            for _ in range(reps):
                plan.step(c0, c1, err, stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            sc = torch.empty(rows, dtype=torch.float32, device=dev); plan.read_scores(sc); torch.cuda.synchronize()
            same = None
            if ref is None: ref = sc.clone()
            else: same = bool(torch.equal(ref, sc))
            print(f"world={world} N={n} E_local={E} {mode:8s}{env}: plan {t_plan*1e3:.0f} ms, sweep {ms:.3f} ms, {E/ms/1e6:.1f} Gedge/s/GPU, blocked={plan.blocked}, same={same}", flush=True)
            plan.close(); del c0, c1
        del off, s, od
        torch.cuda.empty_cache()
main()

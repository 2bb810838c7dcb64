"""scratch: time cz_pagerank_plan_step under the formulations / knobs of csrc/pagerank.hip on the bench graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
from cozo_amd.graph import PageRankPlan

def gen(n, e, device, skew=False):
    g = torch.Generator(device=device); g.manual_seed(4242)
    if skew:
        # power-law-ish: ids = floor(n * u^3) concentrates mass on low ids (hubs)
        dst = (torch.rand(e, generator=g, device=device, dtype=torch.float64) ** 3 * n).to(torch.int64).clamp_(max=n - 1)
        src = (torch.rand(e, generator=g, device=device, dtype=torch.float64) ** 2 * n).to(torch.int64).clamp_(max=n - 1)
    else:
        dst = torch.randint(0, n, (e,), generator=g, device=device, dtype=torch.int64)
        src = torch.randint(0, n, (e,), generator=g, device=device, dtype=torch.int64)
    keep = src != dst
    key = torch.unique(dst[keep] * n + src[keep])
    d = torch.div(key, n, rounding_mode="floor")
    s = (key - d * n).to(torch.int32)
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(torch.bincount(d, minlength=n), 0)
    outdeg = torch.bincount(s.to(torch.int64), minlength=n).to(torch.int32)
    return off.to(torch.int32), s, outdeg, int(off[-1].item())

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    skew = len(sys.argv) > 3 and sys.argv[3] == "skew"
    off, s, od, E = gen(n, e, dev, skew)
    print(f"graph n={n} E={E} skew={skew} maxdeg={int((off[1:].to(torch.int64)-off[:-1].to(torch.int64)).max())}", flush=True)
    stream = torch.cuda.current_stream().cuda_stream
    ref = None
    cfgs = [("gather", {}), ("blocked", {}), ("blocked", {"CZ_PR_SLICE_LOG2": "14"})]
    if os.environ.get("PR_SWEEP") == "chunks":
        cfgs = [("blocked", {})] + [("blocked", {"CZ_PR_CHUNKS": str(c), "CZ_PR_VAL_REUSE": r}) for c in (4, 8, 16, 32) for r in ("0", "1")]
    if os.environ.get("PR_SWEEP") == "xcd":
        cfgs = [("blocked", {"CZ_PR_XCD": "0"}), ("blocked", {"CZ_PR_XCD": "1"}), ("blocked", {"CZ_PR_XCD": "0"}), ("blocked", {"CZ_PR_XCD": "1"})]
    if os.environ.get("PR_SWEEP") == "blocked_only":
        cfgs = [("blocked", {})]
    for mode, env in cfgs:
        for k in ("CZ_PR_CHUNKS", "CZ_PR_SLICE_LOG2", "CZ_PR_VAL_REUSE", "CZ_PR_XCD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan = PageRankPlan(off, s, od, n, 0, n, 0.85, device_ptrs=True, mode=mode)
        torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
        c0 = torch.empty(n, dtype=torch.float32, device=dev); c1 = torch.empty_like(c0)
        err = torch.zeros(1, dtype=torch.float64, device=dev)
        plan.init(c0, stream)
        for _ in range(3):
            plan.step(c0, c1, err, stream); c0, c1 = c1, c0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            plan.step(c0, c1, err, stream); c0, c1 = c1, c0
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        sc = torch.empty(n, dtype=torch.float32, device=dev); plan.read_scores(sc); torch.cuda.synchronize()
        same = None
        if ref is None: ref = sc.clone()
        else: same = bool(torch.equal(ref, sc))
        algo = 4 * E + 4 * (n + 1) + 20 * n
        print(f"{mode:8s} {env}: plan {t_plan*1e3:.1f} ms, step {ms:.3f} ms, {E/ms/1e6:.1f} Gedge/s, roofline {algo/ms/1e6/8000:.3f}, "
              f"scores==gather: {same}, blocked={plan.blocked}", flush=True)
        plan.close(); del c0, c1
main()

"""scratch (round 5): the accumulate formulation of the PageRank sweep against the tile formulation on the bench graphs
(10M / 100M uniform and R-MAT): step time, plan shape, plan build time, scores equal to the tile formulation's after 13 sweeps.
  python scratch/r5_pr.py [uniform|rmat|both] [n] [e]      PR_CFGS=name,name,... picks configurations"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.graph import PageRankPlan

CFGS = {
    "blocked": ("blocked", {}),
    "acc": ("accumulate", {}),
    "acc_b2": ("accumulate", {"CZ_PR_ACC_PER_CU": "2"}),
    "acc_w16": ("accumulate", {"CZ_PR_ACC_WAVES": "16"}),
    "acc_w16b2": ("accumulate", {"CZ_PR_ACC_PER_CU": "2", "CZ_PR_ACC_WAVES": "16"}),
    "acc_a2": ("accumulate", {"CZ_PR_ACC_A_PER_CU": "2"}),
    "acc_a2b2": ("accumulate", {"CZ_PR_ACC_A_PER_CU": "2", "CZ_PR_ACC_PER_CU": "2"}),
    "acc_s512": ("accumulate", {"CZ_PR_ACC_SLICES": "512"}),
    "blocked_c2": ("blocked", {"CZ_PR_CHUNKS": "2"}),
    "blocked_c3": ("blocked", {"CZ_PR_CHUNKS": "3"}),
    "blocked_c4": ("blocked", {"CZ_PR_CHUNKS": "4"}),
    "blocked_c8": ("blocked", {"CZ_PR_CHUNKS": "8"}),
    "auto": (None, {}),
}
KEYS = ("CZ_PR_CHUNKS", "CZ_PR_ACC_PER_CU", "CZ_PR_ACC_WAVES", "CZ_PR_ACC_A_PER_CU", "CZ_PR_ACC_SLICES", "CZ_PR_ACC_GROUPS")

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    kinds = sys.argv[1] if len(sys.argv) > 1 else "both"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    e = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
    names = os.environ.get("PR_CFGS", "blocked,acc,acc_b2,acc_w16,acc_a2,auto").split(",")
    stream = torch.cuda.current_stream().cuda_stream
    args = types.SimpleNamespace()
    for kind in (["uniform", "rmat"] if kinds == "both" else [kinds]):
        off, s, od, max_in = Bn.make_graph(args, torch, None, 0, 1, dev, kind, n, e, 0, n)
        E = int(off[-1].item())
        off32 = off.to(torch.int32)
        print(f"== {kind}: n={n} E={E} longest in-row {max_in}", flush=True)
        ref = None
        for name in names:
            mode, env = CFGS[name]
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                plan = PageRankPlan(off32, s, od, n, 0, n, 0.85, device_ptrs=True, mode=mode)
            except Exception as ex:  # noqa: BLE001
                print(f"{name:10s} plan failed: {ex}", flush=True)
                continue
            torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
            c0 = torch.empty(n, dtype=torch.float32, device=dev); c1 = torch.empty_like(c0)
            err = torch.zeros(1, dtype=torch.float64, device=dev)
            plan.init(c0, stream)
            for _ in range(3):
                plan.step(c0, c1, err, stream); c0, c1 = c1, c0
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    plan.step(c0, c1, err, stream); c0, c1 = c1, c0
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            if hasattr(L, "cz_pagerank_phase_cycles") or os.environ.get("PR_PHASES"):
                import ctypes as C
                try:
                    fn = L.cz_pagerank_phase_cycles
                    fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
                    buf = (C.c_ulonglong * 8)()
                    fn(None, 1)
                    for _ in range(5):
                        plan.step(c0, c1, err, stream); c0, c1 = c1, c0
                    torch.cuda.synchronize()
                    fn(buf, 1)
                    w = max(1, buf[4])
                    print(f"           phases (cycles per wave, {w // 5} waves): " + " ".join(f"{buf[k] / w:.0f}" for k in range(4)), flush=True)
                except AttributeError:
                    pass
            sc = torch.empty(n, dtype=torch.float32, device=dev); plan.read_scores(sc); torch.cuda.synchronize()
            same = None
            if ref is None:
                ref = sc.clone()
            else:
                same = bool(torch.equal(ref, sc))
            algo = 4 * E + 4 * (n + 1) + 20 * n
            print(f"{name:10s} {plan.formulation:10s} step {best:.4f} ms  {E / best / 1e6:7.1f} Gedge/s  frac {algo / best / 1e6 / 8000:.3f}  "
                  f"same_as_first={same}  plan {t_plan * 1e3:.0f} ms  shape {plan.shape}", flush=True)
            plan.close(); del c0, c1, sc
        del off, s, od, off32, ref
        torch.cuda.empty_cache()
main()

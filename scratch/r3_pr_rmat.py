"""scratch (round 3): PageRank sweep time on the bench graphs (uniform / R-MAT, 10M nodes / 100M edges) for settings of
CZ_PR_HEAVY (rows moved behind the others, longest first) and CZ_PR_WAVE_ROW (rows summed by a whole wave, exact_sum.cuh);
scores after 3 sweeps compared with the CPU oracle for the default setting."""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
import bench as Bn
from cozo_amd.graph import PageRankPlan

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
stream = torch.cuda.current_stream().cuda_stream
ap = argparse.ArgumentParser()
ap.add_argument("--kinds", default="rmat,uniform")
ap.add_argument("--parity", type=int, default=1)
ap.add_argument("--only-default", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(multi=False)
n, e = 10_000_000, 100_000_000

def sweep_ms(plan, reps=10):
    cin = torch.empty(n, dtype=torch.float32, device=dev)
    cout = torch.empty_like(cin)
    kerr = torch.zeros(1, dtype=torch.float64, device=dev)
    plan.init(cin, stream)
    plan.step(cin, cout, kerr, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.step(cin, cout, kerr, stream)
        cin, cout = cout, cin
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for kind in a.kinds.split(","):
    off, s, outdeg32, max_in = Bn.make_graph(args, torch, None, 0, 1, dev, kind, n, e, 0, n)
    off32 = off.to(torch.int32)
    print(f"{kind}: {int(off[-1])} edges, longest in-row {max_in}", flush=True)
    settings = [(None, None, {})]
    if kind == "rmat" and not a.only_default:
        settings += [(None, None, {"CZ_PR_XCD_BALANCE": "0"}), (None, None, {"CZ_PR_XCD": "0"}), (None, None, {"CZ_PR_BLOCK_FIXED_COST": "2000"}),
                     (None, None, {"CZ_PR_BLOCK_FIXED_COST": "12000"}), (None, None, {"CZ_PR_BLOCK_FIXED_COST": "20000"}),
                     ("64", None, {}), ("512", None, {}), (None, "1024", {}), (None, "4096", {})]
    for heavy, wrow, extra in settings:
        for k in ("CZ_PR_XCD_BALANCE", "CZ_PR_XCD", "CZ_PR_BLOCK_FIXED_COST"):
            os.environ.pop(k, None)
        os.environ.update(extra)
        for k, v in (("CZ_PR_HEAVY", heavy), ("CZ_PR_WAVE_ROW", wrow)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        t0 = time.time()
        plan = PageRankPlan(off32, s, outdeg32, n, 0, n, 0.85, device_ptrs=True)
        torch.cuda.synchronize()
        tb = time.time() - t0
        ms = sweep_ms(plan)
        algo = 4 * int(off[-1]) + 4 * (n + 1) + 20 * n
        if os.environ.get("COZO_GPU_LIB", "").endswith("prphase.so"):
            import ctypes
            ph = (ctypes.c_ulonglong * 8)()
            L.cz_pagerank_phase_cycles(ph, 1)
            ms = sweep_ms(plan)
            L.cz_pagerank_phase_cycles(ph, 1)
            nb = max(1, ph[4])
            print(f"  phase cycles per workgroup (thread 0): fill {ph[0] / nb:.0f}  queued pieces {ph[1] / nb:.0f}  lane rows {ph[2] / nb:.0f}  "
                  f"wave rows + reduce {ph[3] / nb:.0f}; workgroups {nb / 11:.0f} per sweep, wave rows {ph[5] / 11:.0f}, queued pieces {ph[6] / 11:.0f} per sweep", flush=True)
        print(f"  heavy={heavy or 'default':12s} wave_row={wrow or 'default':8s} {str(extra):38s}: {ms:.4f} ms/sweep  frac {algo / ms / 1e6 / 8000:.4f}  (plan {tb * 1e3:.0f} ms)", flush=True)
        if heavy is None and wrow is None and a.parity:
            from oracle import oracle as O
            from cozo_amd.distributed import ShardedPageRank
            sp = ShardedPageRank(n, 0, 1, dev, lambda c: plan.init(c, stream), lambda ci, co, er: plan.step(ci, co, er, stream))
            sp.run(0.0, 3)
            torch.cuda.synchronize()
            got = plan.read_scores()
            t0 = time.time()
            want, _, _ = O.pagerank(n, off.cpu().numpy().astype(np.uint64), s.cpu().numpy().astype(np.uint32), outdeg32.cpu().numpy().astype(np.uint32), 0.85, 0.0, 3, threads=16)
            print(f"  parity after 3 sweeps ({kind}): {bool(np.array_equal(got, want))}  (oracle {time.time() - t0:.1f}s)", flush=True)
        plan.close()
    del off, s, outdeg32, off32
    torch.cuda.empty_cache()

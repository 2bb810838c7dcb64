"""scratch (round 6): the resident in-place PageRank plan on the bench graph: event-timed sweeps for the urgent gap / graph-replay
settings, parity after 3 sweeps against the oracle.   python scratch/r6_inplace.py [uniform|rmat] [n] [e]   IP_CFGS=name,...  IP_PARITY=0"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.graph import InplacePageRankPlan

CFGS = {
    "t16s16": {},
    "t16s16_gap0": {"CZ_PR_INPLACE_GAP": "0"},
    "t16s16_gap2": {"CZ_PR_INPLACE_GAP": "2"},
    "t16s16_nograph": {"CZ_PR_INPLACE_GRAPH": "0"},
    "t32s32": {"CZ_PR_INPLACE_TILE": "32768", "CZ_PR_INPLACE_SLICE": "32768"},
    "t32s32_p32": {"CZ_PR_INPLACE_TILE": "32768", "CZ_PR_INPLACE_SLICE": "32768", "CZ_PR_INPLACE_PART": "32768"},
    "t16s32": {"CZ_PR_INPLACE_TILE": "16384", "CZ_PR_INPLACE_SLICE": "32768"},
    "t8s16": {"CZ_PR_INPLACE_TILE": "8192", "CZ_PR_INPLACE_SLICE": "16384"},
    "t16s16_p8": {"CZ_PR_INPLACE_PART": "8192"},
    "t16s16_p32": {"CZ_PR_INPLACE_PART": "32768"},
    "gap4": {"CZ_PR_INPLACE_GAP": "4"},
    "gap8": {"CZ_PR_INPLACE_GAP": "8"},
    "gap64": {"CZ_PR_INPLACE_GAP": "64"},
    "t32s32_gap4": {"CZ_PR_INPLACE_TILE": "32768", "CZ_PR_INPLACE_SLICE": "32768", "CZ_PR_INPLACE_GAP": "4"},
    "t16s32_p32": {"CZ_PR_INPLACE_SLICE": "32768", "CZ_PR_INPLACE_PART": "32768"},
    "t16s32_p64": {"CZ_PR_INPLACE_SLICE": "32768", "CZ_PR_INPLACE_PART": "65536"},
    "t16s24_p32": {"CZ_PR_INPLACE_SLICE": "24576", "CZ_PR_INPLACE_PART": "32768"},
    "t32s32_p64": {"CZ_PR_INPLACE_TILE": "32768", "CZ_PR_INPLACE_SLICE": "32768", "CZ_PR_INPLACE_PART": "65536"},
    "jacobi_t16s16": {"AS_JACOBI": "1"},
    "jacobi_t16s32": {"AS_JACOBI": "1", "CZ_PR_INPLACE_SLICE": "32768"},
    "jacobi_t32s32": {"AS_JACOBI": "1", "CZ_PR_INPLACE_TILE": "32768", "CZ_PR_INPLACE_SLICE": "32768"},
    "jacobi_t16s32_p64": {"AS_JACOBI": "1", "CZ_PR_INPLACE_SLICE": "32768", "CZ_PR_INPLACE_PART": "65536"},
    "t32s40": {"CZ_PR_INPLACE_TILE": "32768", "CZ_PR_INPLACE_SLICE": "40448"},
}
KEYS = ("CZ_PR_INPLACE_GAP", "CZ_PR_INPLACE_GRAPH", "CZ_PR_INPLACE_SLICE", "CZ_PR_INPLACE_PART", "CZ_PR_INPLACE_TILE", "AS_JACOBI")

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    e = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
    names = os.environ.get("IP_CFGS", "t16s16,t16s16_gap0,t16s16_gap2,t32s32,t16s32,t8s16,t16s16_p8,t16s16_nograph").split(",")
    stream = torch.cuda.current_stream().cuda_stream
    args = types.SimpleNamespace()
    off, s, od, max_in = Bn.make_graph(args, torch, None, 0, 1, dev, kind, n, e, 0, n)
    E = int(off[-1].item())
    off32 = off.to(torch.int32)
    print(f"== {kind}: n={n} E={E} longest in-row {max_in}", flush=True)
    want = want_j = None
    if os.environ.get("IP_PARITY", "1") != "0":
        from oracle import oracle as O
        t0 = time.time()
        want, _, _ = O.pagerank_mode(n, off.cpu().numpy().astype(np.uint64), s.cpu().numpy().astype(np.uint32), od.cpu().numpy().astype(np.uint32),
                                     0.85, 0.0, 3, mode=O.PR_INPLACE)
        want_j, _, _ = O.pagerank(n, off.cpu().numpy().astype(np.uint64), s.cpu().numpy().astype(np.uint32), od.cpu().numpy().astype(np.uint32), 0.85, 0.0, 3, threads=8)
        print(f"oracle 3 sweeps (both readings): {time.time() - t0:.1f}s", flush=True)
    algo = 4 * E + 4 * (n + 1) + 20 * n
    for name in names:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(CFGS[name])
        t0 = time.perf_counter()
        jac = os.environ.get("AS_JACOBI") == "1"
        plan = InplacePageRankPlan(off32, s, od, 0.85, device_ptrs=True, as_jacobi=jac)
        t_plan = time.perf_counter() - t0
        info = plan.info
        same = None
        if want is not None:
            plan.run(0.0, 3)
            same = bool(np.array_equal(plan.read_scores(), want_j if jac else want))
        plan.init(stream)
        plan.sweeps(3, stream)
        torch.cuda.synchronize()
        if os.environ.get("IP_FEW"):  # counter passes: 2 more sweeps, nothing else (5 sweeps + 1 init launch in all)
            plan.sweeps(2, stream); torch.cuda.synchronize()
            print("few: 5 sweeps", info, flush=True)
            plan.close()
            continue
        times = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.sweeps(10, stream)
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 10)
        best = min(times)
        t0 = time.perf_counter(); it, err = plan.run(0.0, 10); wall = (time.perf_counter() - t0) / 10
        print(f"{name:14s} sweep {best:.4f} ms (runs {', '.join(f'{t:.4f}' for t in times)})  loop {wall * 1e3:.4f} ms/it  {E / best / 1e6:7.1f} Gedge/s  "
              f"frac {algo / best / 1e6 / 8000:.3f}  parity={same}  plan {t_plan:.1f}s  {info}", flush=True)
        plan.close()
main()

"""scratch (round 3): one whole-graph rule on the 10M / 100M bench graph, RUNS times, nothing else on the device --
the process rocprofv3 --pmc wraps to get the HBM traffic of a BFS / an SSSP call (profiles/make_pmc_traffic.py).
    python scratch/r3_rule_runs.py bfs|sssp [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
from cozo_amd import graph as G

rule, runs = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
n, e = 10_000_000, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
keep = src != dst
key = torch.unique(src[keep] * n + dst[keep])
s = torch.div(key, n, rounding_mode="floor"); t = key - s * n
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
ooff, otgt = off.to(torch.int32).cpu().numpy().view(np.uint32), t.to(torch.int32).cpu().numpy().view(np.uint32)
w = (torch.randint(1, 64, (otgt.size,), generator=g, device=dev, dtype=torch.int32).to(torch.float32) / 8).cpu().numpy()
uoff = utgt = None
if rule in ("cc", "tri", "lp"):  # the symmetrised graph, parallel edges kept (bench.py bench_graph_rules), on the device like there
    key2 = torch.sort(torch.cat([key, t * n + s])).values
    s2 = torch.div(key2, n, rounding_mode="floor")
    off2 = torch.zeros(n + 1, dtype=torch.int64, device=dev); off2[1:] = torch.cumsum(torch.bincount(s2, minlength=n), 0)
    uoff, utgt = off2.to(torch.int32).cpu().numpy().view(np.uint32), (key2 - s2 * n).to(torch.int32).cpu().numpy().view(np.uint32)
    del key2, s2, off2
del src, dst, keep, key, s, t, off
torch.cuda.empty_cache()
starts = np.array([0], dtype=np.uint32)
if rule in ("bfs", "sssp"):
    with G.DeviceGraph.acquire((3, 3), ooff, otgt, w if rule == "sssp" else None) as dg:
        for _ in range(runs):
            if rule == "bfs":
                G.bfs(dg, None, starts, want_depth=True)
            else:
                G.sssp(dg, None, None, starts)
            print(rule, "device ms", G.last_timing()[1], flush=True)
else:
    ones = np.ones(utgt.size, dtype=np.float32)
    for _ in range(runs):
        if rule == "cc":
            G.connected_components(uoff, utgt)
        elif rule == "tri":
            G.clustering_coefficients(uoff, utgt, symmetric=True)  # what the rule (and bench.py) passes
        else:
            G.label_propagation(uoff, utgt, ones, 10, symmetric=True)
        print(rule, "device ms", G.last_timing()[1], flush=True)
print("edges", otgt.size, "runs", runs)
if rule == "sssp" and os.environ.get("SWEEP_DELTA"):
    # near-far bucket width: multiples of the mean edge weight (the default)
    mean = float(w.mean())
    with G.DeviceGraph.acquire((3, 4), ooff, otgt, w) as dg:
        for mult in (0.125, 0.25, 0.5, 1.0, 2.0, 4.0):
            os.environ["CZ_SSSP_DELTA"] = repr(mean * mult)
            best = 1e9
            for _ in range(3):
                G.sssp(dg, None, None, starts)
                best = min(best, G.last_timing()[1])
            print(f"delta = {mult:5.3f} x mean weight ({mean * mult:.3f}): device {best:.2f} ms", flush=True)

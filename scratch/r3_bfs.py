"""scratch (round 3): BFS with one pass over the frontier's adjacency per level against the three-pass level of round 2
(CZ_BFS_PASSES=3), on the 10M / 100M uniform graph and on a skewed one (hubs: long stretches of the next frontier);
parents, depths and discovery order compared between the two."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
from cozo_amd import graph as G

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0

def csr(n, src, dst):
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
    s = torch.div(key, n, rounding_mode="floor"); t = key - s * n
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
    return off.to(torch.int32).cpu().numpy().view(np.uint32), t.to(torch.int32).cpu().numpy().view(np.uint32)

n, e = 10_000_000, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
graphs = {}
graphs["uniform"] = csr(n, torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64), torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64))
u = torch.rand(e, generator=g, device=dev, dtype=torch.float64)
graphs["skewed (sources ~ u^3)"] = csr(n, (u ** 3 * n).to(torch.int64).clamp_(max=n - 1), torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64))
torch.cuda.empty_cache()
starts = np.array([0], dtype=np.uint32)
for name, (off, tgt) in graphs.items():
    print(f"{name}: {tgt.size} edges, max out-degree {int(np.diff(off.astype(np.int64)).max())}", flush=True)
    res = {}
    for mode in ("3", "1", "3", "1"):
        os.environ["CZ_BFS_PASSES"] = mode
        G.bfs(off, tgt, starts, want_depth=True, want_order=True)
        t0 = time.perf_counter()
        r = G.bfs(off, tgt, starts, want_depth=True, want_order=True)
        dt = time.perf_counter() - t0
        up, dev_ms, down = G.last_timing()
        print(f"  passes={mode}: wall {dt * 1e3:7.1f} ms  upload {up:6.1f}  device {dev_ms:7.2f} ms ({tgt.size / dev_ms / 1e6:6.2f} G edges/s)  download {down:5.1f}  reached {int(r[3][0])}", flush=True)
        res[mode] = r
    same = all(np.array_equal(a, b) for a, b in zip(res["3"][:3], res["1"][:3])) and np.array_equal(res["3"][3], res["1"][3])
    print(f"  parents, depths, discovery order identical between the two levels: {same}", flush=True)
os.environ.pop("CZ_BFS_PASSES", None)
# repeated call on a held graph, result arrays reused
off, tgt = graphs["uniform"]
out = {}
with G.DeviceGraph.acquire((9, 9), off, tgt, None) as dg:
    G.bfs(dg, None, starts, want_depth=True, out=out)
for _ in range(3):
    t0 = time.perf_counter()
    with G.DeviceGraph.acquire((9, 9), off, tgt, None) as dg:
        G.bfs(dg, None, starts, want_depth=True, out=out)
    print(f"  held graph, reused result arrays: wall {(time.perf_counter() - t0) * 1e3:.2f} ms  (device {G.last_timing()[1]:.2f}, download {G.last_timing()[2]:.2f})", flush=True)

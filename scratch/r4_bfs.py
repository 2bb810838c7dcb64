"""scratch (round 4): device time of BFS / SSSP / CC on the resident 10M / 100M bench graph, several calls (COZO_GPU_LIB picks the library)"""
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
n, e = 10_000_000, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
keep = src != dst
key = torch.unique(src[keep] * n + dst[keep])
s = torch.div(key, n, rounding_mode="floor"); t = key - s * n
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
ooff, otgt = off.to(torch.int32).cpu().numpy().view(np.uint32), t.to(torch.int32).cpu().numpy().view(np.uint32)
del src, dst, keep, key, s, t, off
torch.cuda.empty_cache()
starts = np.array([0], dtype=np.uint32)
out = {}
import hashlib
with G.DeviceGraph.acquire((3, 3), ooff, otgt, None) as dg:
    for i in range(5):
        par, dep, order, _ = G.bfs(dg, None, starts, want_depth=True, want_order=True, out=out)
        print("bfs device ms", round(G.last_timing()[1], 3), hashlib.sha1(par.tobytes() + dep.tobytes() + order.tobytes()).hexdigest()[:12], flush=True)

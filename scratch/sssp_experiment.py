"""scratch: one big relaxation round of the single-source SSSP on the 10M / 100M graph replayed with pieces of the kernel left
out (CZ_SSSP_EXPERIMENT=1, csrc/graph.hip SsspBatch::experiment) -- prints to stderr"""
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
w = (torch.randint(1, 64, (otgt.size,), generator=g, device=dev, dtype=torch.int32).to(torch.float32) / 8).cpu().numpy()
del src, dst, keep, key, s, t, off
torch.cuda.empty_cache()
t0 = time.perf_counter()
dist, _ = G.sssp(ooff, otgt, w, np.array([0], dtype=np.uint32))
print("wall", time.perf_counter() - t0, "reached", int(np.isfinite(dist[0]).sum()), flush=True)

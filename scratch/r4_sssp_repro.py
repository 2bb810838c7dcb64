"""scratch (round 4): the held cz_sssp_on call whose FIRST kernel (fill_u64_kernel over 80 MB of pool memory) takes 30-39 ms inside
the whole bench (gpurun_out/r4f: `run: fill dp +30246 us`) and 39 us in a fresh process.  Hypothesis: the driver's pending
page-table work after tens of GB were freed.  Here: the held sequence fresh, then again after 60 GB were written and freed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CZ_SSSP_TRACE"] = "1"
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
starts = np.array([0], dtype=np.uint32)


def held(tag, key):
    for i in range(3):
        t0 = time.perf_counter()
        with G.DeviceGraph.acquire(key, ooff, otgt, w) as dg:
            G.sssp(dg, None, None, starts)
        up, dv, dn = G.last_timing()
        print(f"== {tag} call {i}: wall {1e3 * (time.perf_counter() - t0):6.1f} ms  device lap {dv:5.1f}", file=sys.stderr, flush=True)


held("fresh process", (0xC0, 3))
for gb, how in ((60, "zeros"), (120, "zeros")):
    x = [torch.zeros(gb // 4 * (1 << 30), dtype=torch.uint8, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
    held(f"after {gb} GB were written and freed (torch), graph still held", (0xC0, 3))
    L.cz_graph_cache_clear()
    held(f"after {gb} GB, graph uploaded anew", (0xC0, 4 + gb))

"""scratch (round 6): clustering coefficients on the bench's R-MAT / uniform graph: device ms per (CZ_TRI_MERGE, CZ_TRI_SCAN) setting,
every setting's counts compared with the first one's and a node sample with the oracle.  python scratch/r6_tri.py rmat|uniform"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cozo_amd import _lib, graph as G
import bench as Bn
from oracle import oracle as O
O.build()
L = _lib.lib()
assert L.cz_init(0) == 0
kind = sys.argv[1] if len(sys.argv) > 1 else "rmat"
n, e = 10_000_000, 100_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
if kind == "uniform":
    src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
    dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
else:
    scale = (n - 1).bit_length()
    src, dst = Bn.rmat_edges(torch, scale, int(e * 1.6), dev, 4243)
    keep = (src < n) & (dst < n) & (src != dst)
    key = torch.unique(src[keep] * n + dst[keep])
    if key.numel() > e:
        sel = torch.randperm(key.numel(), generator=g, device=dev)[:e]
        key = torch.sort(key[sel]).values
sN = torch.div(key, n, rounding_mode="floor"); t = key - sN * n
key2 = torch.sort(torch.cat([key, t * n + sN])).values
s2 = torch.div(key2, n, rounding_mode="floor"); t2 = key2 - s2 * n
off2 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
off2[1:] = torch.cumsum(torch.bincount(s2, minlength=n), 0)
uoff, utgt = off2.to(torch.int32).cpu().numpy().view(np.uint32), t2.to(torch.int32).cpu().numpy().view(np.uint32)
print(kind, "max degree", int(np.diff(uoff.astype(np.int64)).max()), flush=True)
o64 = uoff.astype(np.int64)
row = np.repeat(np.arange(n, dtype=np.int64), np.diff(o64))
above = utgt.astype(np.int64) > row
dplus = np.bincount(row[above], minlength=n)
work = np.bincount(row[above], weights=dplus[utgt[above]].astype(np.float64), minlength=n)
for thr in (32, 64, 256, 1024, 16384):
    sel = dplus > thr
    print(f"m > {thr}: {int(sel.sum())} nodes, work {work[sel].sum():.3e}, max {work[sel].max() if sel.any() else 0:.3e}", flush=True)
top = np.argsort(-work)[:8]
print("top work", [(int(v), int(dplus[v]), float(work[v])) for v in top], flush=True)
print("sum d+^2 (pair form)", float((dplus.astype(np.float64) ** 2).sum()), flush=True)
del row, above
first = None
for merge, scan in ((64, 128), (32, 64), (48, 128)):
    os.environ["CZ_TRI_MERGE"], os.environ["CZ_TRI_SCAN"] = str(merge), str(scan)
    best = 1e9
    for _ in range(2):
        tri, deg = G.clustering_coefficients(uoff, utgt, symmetric=True)
        best = min(best, G.last_timing()[1])
    if first is None:
        first = tri.copy()
    print(f"merge {merge} scan {scan}: device {best:.1f} ms  same_as_first {np.array_equal(tri, first)}  total {int(tri.sum())}", flush=True)
nodes, otri, ne = O.clustering_coefficients_sample(n, uoff, utgt, first=37 if kind == 'rmat' else 0, step=64 if kind == 'rmat' else 16, max_seconds=15.0)
print("oracle sample", len(nodes), "nodes, equal:", np.array_equal(first[nodes], otri), flush=True)

"""scratch: the other whole-graph rules on a BASELINE-sized graph (10M nodes / 100M edges, uniform): wall time of the C ABI call
(host pointers: H2D + kernels + D2H) -- run it under `rocprofv3 --kernel-trace --stats` for the kernel-only split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
from cozo_amd import graph as G

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, e = int(os.environ.get("GN", 10_000_000)), int(os.environ.get("GE", 100_000_000))
    g = torch.Generator(device=dev); g.manual_seed(7)
    src = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
    dst = torch.randint(0, n, (e,), generator=g, device=dev, dtype=torch.int64)
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])          # directed out-CSR, CsrLayout::Sorted
    s = torch.div(key, n, rounding_mode="floor"); t = key - s * n
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
    ooff, otgt = off.to(torch.int32).cpu().numpy().view(np.uint32), t.to(torch.int32).cpu().numpy().view(np.uint32)
    E = otgt.size
    w = (torch.randint(1, 64, (E,), generator=g, device=dev, dtype=torch.int32).to(torch.float32) / 8).cpu().numpy()
    # symmetrised graph for CC / triangles (parallel edges kept, like as_directed_graph(undirected = true))
    key2 = torch.sort(torch.cat([key, t * n + s])).values
    s2 = torch.div(key2, n, rounding_mode="floor"); t2 = key2 - s2 * n
    off2 = torch.zeros(n + 1, dtype=torch.int64, device=dev); off2[1:] = torch.cumsum(torch.bincount(s2, minlength=n), 0)
    uoff, utgt = off2.to(torch.int32).cpu().numpy().view(np.uint32), t2.to(torch.int32).cpu().numpy().view(np.uint32)
    del src, dst, keep, key, key2, s, t, s2, t2, off, off2
    torch.cuda.empty_cache()
    print(f"graph: {n} nodes, {E} directed edges ({utgt.size} symmetrised)", flush=True)

    def timed(name, fn, edges, note=""):
        fn()  # warm (allocations, code objects)
        t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
        up, dev_ms, down = G.last_timing()
        print(f"{name:34s} {dt * 1e3:9.1f} ms wall  {edges / dt / 1e9:7.2f} G edges/s   upload {up:6.1f}  device {dev_ms:7.2f} ms "
              f"({edges / dev_ms / 1e6:7.2f} G edges/s)  download {down:5.1f}  {note}", flush=True)
        return r
    starts = np.array([0], dtype=np.uint32)
    par, dep, _, _ = timed("cz_bfs (1 start, full traversal)", lambda: G.bfs(ooff, otgt, starts, want_depth=True), E)
    reached = int((dep[0] != 0xFFFFFFFF).sum()); print(f"    reached {reached} nodes, depth {int(dep[0][dep[0] != 0xFFFFFFFF].max())}")
    grp, k = timed("cz_connected_components", lambda: G.connected_components(uoff, utgt), utgt.size)
    print(f"    {k} components")
    dist, _ = timed("cz_sssp (1 start)", lambda: G.sssp(ooff, otgt, w, starts), E)
    print(f"    reached {int(np.isfinite(dist[0]).sum())} nodes, max cost {float(dist[0][np.isfinite(dist[0])].max()):.3f}")
    for name, key, a, b, c, fn in (("cz_bfs_on (graph held by the cache)", (1, 1), ooff, otgt, None, lambda dg: G.bfs(dg, None, starts, want_depth=True)),
                                  ("cz_connected_components_on (held)", (1, 2), uoff, utgt, None, lambda dg: G.connected_components(dg)),
                                  ("cz_sssp_on (held)", (1, 3), ooff, otgt, w, lambda dg: G.sssp(dg, None, None, starts))):
        def call():
            with G.DeviceGraph.acquire(key, a, b, c) as dg:
                return fn(dg)
        timed(name, call, b.size)
    L.cz_graph_cache_clear()
    tri, deg = timed("cz_clustering_coefficients (symmetry verified exactly)", lambda: G.clustering_coefficients(uoff, utgt), utgt.size)
    tri, deg = timed("cz_clustering_coefficients (CZ_ADJ_SYMMETRIC)", lambda: G.clustering_coefficients(uoff, utgt, symmetric=True), utgt.size)
    print(f"    {int(tri.sum())} (node, triangle) incidences, max degree {int(deg.max())}")
    if os.environ.get("WITH_LP"):
        ones = np.ones(utgt.size, dtype=np.float32)
        lab, it, k = timed("cz_label_propagation (<= 10 iter; symmetry verified exactly)", lambda: G.label_propagation(uoff, utgt, ones, 10), utgt.size * 10)
        lab, it, k = timed("cz_label_propagation (<= 10 iter; CZ_ADJ_SYMMETRIC)", lambda: G.label_propagation(uoff, utgt, ones, 10, symmetric=True), utgt.size * 10)
        print(f"    {it} iterations over {k} colour classes, {len(np.unique(lab))} labels left")
def all_sources():
    """the device part of ClosenessCentrality / BetweennessCentrality: cz_sssp from EVERY node, 256 starts per call"""
    n, e = int(os.environ.get("AN", 20_000)), int(os.environ.get("AE", 200_000))
    rng = np.random.default_rng(3)
    key = np.unique(rng.integers(0, n, e, dtype=np.int64) * n + rng.integers(0, n, e, dtype=np.int64))
    s, t = key // n, key % n
    keep = s != t
    s, t = s[keep], t[keep]
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum(np.bincount(s, minlength=n))
    tgt = t.astype(np.uint32)
    w = (rng.integers(1, 64, tgt.size) / 8).astype(np.float32)
    G.sssp(off, tgt, w, np.arange(256, dtype=np.uint32))  # warm
    t0 = time.perf_counter()
    for b0 in range(0, n, 256):
        G.sssp(off, tgt, w, np.arange(b0, min(n, b0 + 256), dtype=np.uint32))
    dt = time.perf_counter() - t0
    print(f"all-sources cz_sssp: {n} starts on {n} nodes / {tgt.size} edges in batches of 256: {dt:.2f} s "
          f"({n * tgt.size / dt / 1e9:.2f} G edge relaxations-equivalent/s, {dt / n * 1e3:.3f} ms per start)", flush=True)
    t0 = time.perf_counter()
    cc = G.closeness(off, tgt, w)
    dt = time.perf_counter() - t0
    print(f"cz_closeness: {n} nodes / {tgt.size} edges: {dt:.2f} s (device {G.last_timing()[1] / 1e3:.2f} s), max {np.nanmax(cc[np.isfinite(cc)]):.4f}", flush=True)
    G.betweenness(off[:2001].copy(), tgt[:off[2000]] % 2000, w[:off[2000]])  # warm
    t0 = time.perf_counter()
    c = G.betweenness(off, tgt, w)
    dt = time.perf_counter() - t0
    print(f"cz_betweenness: {n} nodes / {tgt.size} edges: {dt:.2f} s ({dt / n * 1e3:.3f} ms per source; device {G.last_timing()[1] / 1e3:.2f} s), "
          f"max centrality {c.max():.1f}", flush=True)


if not os.environ.get("ONLY_ALL_SOURCES"):
    main()
all_sources()

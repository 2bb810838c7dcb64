"""Freezes what the oracle (oracle/cozo_oracle.c) computes TODAY on seeded inputs into tests/golden/oracle_freeze.json: sha256
digests of its outputs per function.  The reference holds no numeric goldens for this path (parity unpinned),
so this does not pin the oracle to the reference -- it pins it to ITSELF: a later edit of the restatement that changes any result
(a summation order, a tie-break, an id order) fails tests/test_oracle_freeze.py and has to be justified against the cited reference
lines before the digests are regenerated with this script.

    python tests/golden/make_oracle_freeze.py            # rewrites oracle_freeze.json
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _h(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()[:32]


def compute():
    from oracle import oracle as O
    from tests import util
    out = {}
    rng = np.random.default_rng(1234)
    # ---- distances (hnsw.rs:66-109), both summation orders, the dimensions of SURVEY 8c(i)
    for d in (1, 2, 7, 8, 9, 128, 768, 1536):
        a = rng.standard_normal((16, d)).astype(np.float32)
        b = rng.standard_normal((16, d)).astype(np.float32)
        pairs = np.stack([np.arange(16, dtype=np.uint32), np.arange(16, dtype=np.uint32)[::-1]], 1).copy()
        for metric in (O.L2, O.COSINE, O.IP):
            for mode in (O.DOT_NDARRAY, O.DOT_GPU):
                out[f"distance d={d} metric={metric} mode={mode}"] = _h(O.distance_pairs(metric, b, a, pairs, mode))
    # ---- HNSW: configs[0]-shaped (smaller), build + search (hnsw.rs:155-587, 869-1012)
    x = np.random.default_rng(42).random((2000, 32), dtype=np.float32)
    q = np.random.default_rng(43).random((64, 32), dtype=np.float32)
    for metric, name in ((O.L2, "L2"), (O.COSINE, "Cosine"), (O.IP, "IP")):
        _, flat = util.build_index(O, x, metric, 8, 40)
        out[f"hnsw build {name} tables"] = _h(*[np.asarray(t) for t in flat.level_nbrs], *[np.asarray(t) for t in flat.level_nodes if t is not None])
        for ef in (10, 40):
            ids, dist, cnt, nd = flat.knn_batch(q, 10, ef)
            out[f"hnsw knn {name} ef={ef}"] = _h(ids, dist, cnt, np.array([nd]))
        ids, dist, cnt, nd = flat.knn_batch(q, 10, 40, radius=float(np.median(dist)))
        out[f"hnsw knn {name} radius"] = _h(ids, dist, cnt)
    # ---- index construction with the two options and over rows that carry several vectors (hnsw.rs:499-511, 524-536, 609-610),
    # removal (hnsw.rs:754-868): tables, degrees of the self rows, soft-deleted and dangling rows
    xs, lv = x[:400], O.random_levels(400, 4, 3)
    row_of = np.sort(np.random.default_rng(8).integers(0, 150, 400)).astype(np.uint32)
    for tag, kw, rows in (("extend", dict(extend_candidates=True), None), ("extend+keep", dict(extend_candidates=True, keep_pruned_connections=True), None),
                          ("rows", {}, row_of), ("rows+extend", dict(extend_candidates=True), row_of)):
        b = O.HnswBuilder(32, O.L2, 4, 16, **kw)
        if rows is not None:
            b.set_row_of(rows)
        b.insert(xs, lv)
        flat = b.export()
        deg = [np.array([b.degree(int(v), l) for v in flat.level_nodes[l]]) for l in range(flat.n_levels)]
        out[f"hnsw build {tag}"] = _h(*flat.level_nbrs, *deg, np.array([flat.entry, b.link_rows(True), b.link_rows()]))
        b.remove(range(0, 400, 7))
        flat = b.export()
        deg = [np.array([b.degree(int(v), l) for v in flat.level_nodes[l]]) for l in range(flat.n_levels)]
        out[f"hnsw remove after {tag}"] = _h(*flat.level_nbrs, *flat.level_nodes, *deg, np.array([flat.entry, b.dangling_links()]))
    # ---- graph rules on one seeded relation
    frm, to = util.random_relation(3000, 14000, 77)
    w = (np.random.default_rng(5).integers(0, 40, len(frm)) / 8).astype(np.float32)
    g = util.graph_from_relation(O, frm, to)
    u = util.graph_from_relation(O, frm, to, undirected=True)
    gw = util.graph_from_relation(O, frm, to, weights=w)
    out["assign_ids + csr"] = _h(g["fi"], g["ti"], g["ooff"], g["otgt"], g["ioff"], g["isrc"])
    out["weighted csr"] = _h(gw["ooff"], gw["otgt"], gw["ow"])
    for damping, tol, it in ((0.85, 1e-4, 10), (0.5, 0.0, 4)):
        s, iters, err = O.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], damping, tol, it)
        out[f"pagerank d={damping} tol={tol} it={it}"] = _h(s, np.array([iters]), np.array([err]))
    order, parent, _ = O.bfs_order(g["n"], g["ooff"], g["otgt"], 0)
    out["bfs order + parents"] = _h(order, parent)
    out["shortest_path_bfs"] = _h(O.shortest_path_bfs(g["n"], g["ooff"], g["otgt"], 3, np.array([5, 17, 2999], dtype=np.uint32)))
    grp, k = O.tarjan_groups(u["n"], u["ooff"], u["otgt"])
    out["connected components"] = _h(grp, np.array([k]))
    cc, tri, deg = O.clustering_coefficients(u["n"], u["ooff"], u["otgt"])
    out["clustering coefficients"] = _h(cc, tri, deg)
    d, p = O.dijkstra(gw["n"], gw["ooff"], gw["otgt"], gw["ow"], 7)
    out["dijkstra costs"] = _h(d)
    small_f, small_t = util.random_relation(60, 220, 9)
    gs = util.graph_from_relation(O, small_f, small_t, weights=np.random.default_rng(9).integers(1, 4, len(small_f)).astype(np.float32))
    out["betweenness (literal enumeration)"] = _h(O.betweenness(gs["n"], gs["ooff"], gs["otgt"], gs["ow"]))
    colour, kc = O.lp_colouring(gw["n"], gw["ooff"], gw["otgt"])
    labels, it = O.label_propagation(gw["n"], gw["ooff"], gw["otgt"], gw["ow"], 10)
    out["label propagation (fixed order)"] = _h(colour, np.array([kc]), labels, np.array([it]))
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_freeze.json")
    with open(path, "w") as f:
        json.dump(compute(), f, indent=1, sort_keys=True)
    print("wrote", path)

#!/usr/bin/env python
"""Builds tests/golden/air_routes.npz + air_routes_expect.json from the reference's own graph fixture.

Run in the authoring container (the fixture does not travel to the GPU box):

    python tests/golden/make_air_routes_golden.py /root/reference/cozo-core/tests

Input: air-routes-latest-{nodes,edges}.csv (Kelvin Lawrence's public air-routes data set, loaded by
cozo-core/tests/air_routes.rs:33-140 into `airport{code}` and `route{fr, to => dist}`).  The .npz holds the `route`
relation in compact form (airport codes, endpoints as indices, integer mile distances).  The expectations are
computed with scipy.sparse.csgraph / float64 numpy -- implementations that share no code with oracle/ or the HIP
kernels -- so they pin the oracle on the reference's fixture:
  * route count (air_routes.rs:208 asserts 50637 * 5),
  * BFS PEK -> LHR (air_routes.rs:212-236 asserts the endpoints): hop count,
  * connected components of the symmetrised graph (air_routes.rs:254-267 runs it): partition,
  * Dijkstra JFK -> all (air_routes.rs:300-316 runs JFK -> KUL): exact costs (integer miles < 2^24, exact in f32),
  * PageRank, 10 iterations, theta 0.85, in float64 with the graph crate's semantics (no dangling redistribution).
"""
import csv
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
from scipy.sparse import csgraph


def main(src):
    here = os.path.dirname(os.path.abspath(__file__))
    idx2code, airports = {}, []
    with open(os.path.join(src, "air-routes-latest-nodes.csv"), newline="") as f:
        r = csv.reader(f)
        next(r)
        for row in r:
            idx2code[int(row[0])] = row[3]
            if row[1] == "airport":
                airports.append(row[3])
    fr, to, dist = [], [], []
    with open(os.path.join(src, "air-routes-latest-edges.csv"), newline="") as f:
        r = csv.reader(f)
        next(r)
        for row in r:
            if row[3] == "route":
                fr.append(idx2code[int(row[1])])
                to.append(idx2code[int(row[2])])
                dist.append(int(row[4]))
    codes = sorted(set(airports))
    pos = {c: i for i, c in enumerate(codes)}
    fi = np.array([pos[c] for c in fr], dtype=np.uint16)
    ti = np.array([pos[c] for c in to], dtype=np.uint16)
    di = np.array(dist, dtype=np.uint16)
    assert max(dist) < 65536 and len(codes) < 65536
    # `route` is keyed by (fr, to): a set
    assert len({(a, b) for a, b in zip(fr, to)}) == len(fr)
    np.savez_compressed(os.path.join(here, "air_routes.npz"), codes=np.array(codes), fr=fi, to=ti, dist=di)

    n = len(codes)
    A = sp.csr_matrix((np.ones(len(fi)), (fi.astype(np.int64), ti.astype(np.int64))), shape=(n, n))
    W = sp.csr_matrix((di.astype(np.float64), (fi.astype(np.int64), ti.astype(np.int64))), shape=(n, n))
    hops = csgraph.shortest_path(A, method="D", unweighted=True, indices=[pos["PEK"]])[0]
    ncc, labels = csgraph.connected_components(A, directed=False)
    in_graph = np.zeros(n, dtype=bool)
    in_graph[fi] = True
    in_graph[ti] = True
    costs = csgraph.dijkstra(W, directed=True, indices=[pos["JFK"]])[0]
    # PageRank in float64 over the nodes that appear in `route` (graph::page_rank semantics)
    used = np.flatnonzero(in_graph)
    remap = -np.ones(n, dtype=np.int64)
    remap[used] = np.arange(used.size)
    m = used.size
    a = remap[fi.astype(np.int64)]
    b = remap[ti.astype(np.int64)]
    outdeg = np.bincount(a, minlength=m).astype(np.float64)
    score = np.full(m, 1.0 / m)
    M = sp.csr_matrix((np.ones(a.size), (b, a)), shape=(m, m))
    for _ in range(10):
        with np.errstate(divide="ignore"):
            contrib = score / outdeg
        contrib[outdeg == 0] = 0.0  # never read: a sink is nobody's in-neighbour
        score = (1.0 - 0.85) / m + 0.85 * (M @ contrib)
    expect = {
        "_made_by": "tests/golden/make_air_routes_golden.py (scipy.sparse.csgraph + float64 numpy)",
        "routes": int(len(fi)), "airports": n, "nodes_in_routes": int(m),
        "bfs": {"from": "PEK", "to": "LHR", "hops": int(hops[pos["LHR"]])},
        "bfs_hops_from_PEK": {c: (None if not np.isfinite(hops[pos[c]]) else int(hops[pos[c]]))
                              for c in ("LHR", "JFK", "SIN", "ANC", "SYD", "KUL", "PEK")},
        "cc": {"components_among_route_nodes": int(len(set(labels[in_graph].tolist()))),
               "largest": int(np.bincount(labels[in_graph]).max()),
               "same_component": [["PEK", "LHR", bool(labels[pos["PEK"]] == labels[pos["LHR"]])]]},
        "dijkstra": {"from": "JFK", "costs": {c: (None if not np.isfinite(costs[pos[c]]) else float(costs[pos[c]]))
                                              for c in ("KUL", "LHR", "SYD", "PEK", "ANC", "JFK")},
                     "sum_finite": float(costs[np.isfinite(costs) & in_graph].sum()),
                     "reachable": int((np.isfinite(costs) & in_graph).sum())},
        "pagerank": {"iterations": 10, "theta": 0.85,
                     "top10": [[codes[used[i]], float(score[i])] for i in np.argsort(-score)[:10]],
                     "sum": float(score.sum())},
    }
    # full-vector goldens, compact
    np.savez_compressed(os.path.join(here, "air_routes_expect.npz"), cc_label=labels.astype(np.int32),
                        in_graph=in_graph, dijkstra_jfk=costs, hops_pek=hops, pagerank_nodes=used.astype(np.int32),
                        pagerank_f64=score)
    with open(os.path.join(here, "air_routes_expect.json"), "w") as f:
        json.dump(expect, f, indent=1)
    print(json.dumps(expect, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/cozo-core/tests")

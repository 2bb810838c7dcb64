"""The inputs of oracle/ref_fixtures (the real cozo-core run by oracle/ref_fixtures/make_ref_fixtures.sh): a list of
CozoScript steps over seeded data.  tests/test_ref_fixtures.py regenerates the same data from `dataset()` and compares
what the reference returned (tests/golden/ref_fixtures.json, when a box with cargo has produced it) with the oracle.

    python tests/golden/make_ref_inputs.py inputs.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

DIMS = (1, 2, 7, 8, 9, 128, 768)
HNSW = dict(n=600, dim=16, m=8, ef_construction=40, queries=32, k=10, ef=40)
METRICS = ("L2", "Cosine", "IP")
# round 3: the two paths that were added to the device build -- small, so that the literal row store of the tests can stand in
EXT = dict(n=150, dim=8, m=3, ef_construction=10, queries=16, k=6, ef=20)   # extend_candidates: true
ROWS = dict(rows=60, dim=8, m=3, ef_construction=10, queries=16, k=6, ef=20)  # a List of 1..3 vectors per row


def dataset():
    """everything seeded, as numpy arrays"""
    rng = np.random.default_rng(20260923)
    d = {"pairs": {}}
    for dim in DIMS:
        a = rng.standard_normal((12, dim)).astype(np.float32)
        b = rng.standard_normal((12, dim)).astype(np.float32)
        if dim >= 2:  # the known answers of runtime/tests.rs:691-697 ride along: [1,2].[2,3], v against itself, a zero vector
            a[0, :2], b[0, :2] = (1.0, 2.0), (2.0, 3.0)
            a[0, 2:], b[0, 2:] = 0.0, 0.0
            b[1] = a[1]
            a[2] = 0.0
        d["pairs"][dim] = (a, b)
    d["vectors"] = np.random.default_rng(42).random((HNSW["n"], HNSW["dim"]), dtype=np.float32)
    d["queries"] = np.random.default_rng(43).random((HNSW["queries"], HNSW["dim"]), dtype=np.float32)
    d["ext_vectors"] = np.random.default_rng(44).random((EXT["n"], EXT["dim"]), dtype=np.float32)
    d["ext_queries"] = np.random.default_rng(45).random((EXT["queries"], EXT["dim"]), dtype=np.float32)
    rr = np.random.default_rng(46)
    centres = rr.random((ROWS["rows"], ROWS["dim"]), dtype=np.float32)
    d["rows_vectors"] = [[(centres[i] + 0.05 * rr.standard_normal(ROWS["dim"])).astype(np.float32) for _ in range(1 + i % 3)]
                         for i in range(ROWS["rows"])]  # a row's vectors are each other's nearest
    d["rows_queries"] = rr.random((ROWS["queries"], ROWS["dim"]), dtype=np.float32)
    from tests import util
    frm, to = util.random_relation(3000, 14000, 77)
    d["frm"], d["to"] = frm, to
    d["w"] = (np.random.default_rng(5).integers(0, 40, len(frm)) / 8).astype(np.float32)
    # a graph of several 16 384-node chunks (graph crate's scheduler unit): where a thread-schedule dependence would show
    d["big_frm"], d["big_to"] = util.random_relation(40000, 240000, 78)
    d["dijkstra_start"] = int(frm[7])
    d["bfs_start"] = int(frm[3])
    d["bfs_goals"] = [int(to[5]), int(to[17]), int(frm[-1])]
    return d


def f32_list(x):
    return [float(v) for v in np.asarray(x, dtype=np.float32)]  # an f32 is exact as an f64 JSON number


def steps():
    d = dataset()
    out = []
    for dim in DIMS:
        a, b = d["pairs"][dim]
        out.append(dict(name=f"distances d={dim}",
                        script="rows[i, a, b] <- $rows\n"
                               "?[i, l2, ip, cos] := rows[i, a, b], va = vec(a), vb = vec(b), l2 = l2_dist(va, vb), ip = ip_dist(va, vb), cos = cos_dist(va, vb)\n"
                               ":order i",
                        params=dict(rows=[[i, f32_list(a[i]), f32_list(b[i])] for i in range(len(a))])))
    vec_rows = [[i, f32_list(v)] for i, v in enumerate(d["vectors"])]
    q_rows = [[i, f32_list(v)] for i, v in enumerate(d["queries"])]
    for metric in METRICS:
        t = f"vt_{metric.lower()}"
        out.append(dict(name=f"hnsw create table {metric}", mutable=True, script=f":create {t} {{k: Int => v: <F32; {HNSW['dim']}>}}"))
        out.append(dict(name=f"hnsw put {metric}", mutable=True, script=f"?[k, v] <- $rows\n:put {t} {{k => v}}", params=dict(rows=vec_rows)))
        out.append(dict(name=f"hnsw create index {metric}", mutable=True,
                        script=f"::hnsw create {t}:idx {{dim: {HNSW['dim']}, m: {HNSW['m']}, dtype: F32, fields: [v], distance: {metric}, "
                               f"ef_construction: {HNSW['ef_construction']}}}"))
        out.append(dict(name=f"hnsw index rows {metric}",
                        script=f"?[layer, fr_k, fr__field, fr__sub_idx, to_k, to__field, to__sub_idx, dist, ignore_link] := "
                               f"*{t}:idx{{layer, fr_k, fr__field, fr__sub_idx, to_k, to__field, to__sub_idx, dist, ignore_link}}"))
        out.append(dict(name=f"hnsw knn {metric}",
                        script=f"qs[qi, qv] <- $queries\n"
                               f"?[qi, k, dist] := qs[qi, qv], q = vec(qv), ~{t}:idx{{k | query: q, k: {HNSW['k']}, ef: {HNSW['ef']}, bind_distance: dist}}\n"
                               f":order qi, dist, k",
                        params=dict(queries=q_rows)))
    # ---- extend_candidates (hnsw.rs:499-511): the index rows WITH the hash column -- a shrink that selects its own target writes a
    # link row onto the self row and hnsw_put_vector puts the self row back (:413-433, :352-357)
    all_cols = "layer, fr_k, fr__field, fr__sub_idx, to_k, to__field, to__sub_idx, dist, hash, ignore_link"
    out.append(dict(name="ext create table", mutable=True, script=f":create vt_ext {{k: Int => v: <F32; {EXT['dim']}>}}"))
    out.append(dict(name="ext put", mutable=True, script="?[k, v] <- $rows\n:put vt_ext {k => v}",
                    params=dict(rows=[[i, f32_list(v)] for i, v in enumerate(d["ext_vectors"])])))
    out.append(dict(name="ext create index", mutable=True,
                    script=f"::hnsw create vt_ext:idx {{dim: {EXT['dim']}, m: {EXT['m']}, dtype: F32, fields: [v], distance: L2, "
                           f"ef_construction: {EXT['ef_construction']}, extend_candidates: true}}"))
    out.append(dict(name="ext index rows", script=f"?[{all_cols}] := *vt_ext:idx{{{all_cols}}}"))
    out.append(dict(name="ext knn",
                    script=f"qs[qi, qv] <- $queries\n"
                           f"?[qi, k, dist] := qs[qi, qv], q = vec(qv), ~vt_ext:idx{{k | query: q, k: {EXT['k']}, ef: {EXT['ef']}, bind_distance: dist}}\n"
                           f":order qi, dist, k",
                    params=dict(queries=[[i, f32_list(v)] for i, v in enumerate(d["ext_queries"])])))
    # ---- rows that carry several vectors (hnsw.rs:694-706): links inside a base row are stored and never read (:609-610)
    out.append(dict(name="rows create table", mutable=True, script=f":create vt_rows {{k: Int => vs: [<F32; {ROWS['dim']}>]}}"))
    out.append(dict(name="rows put", mutable=True, script="?[k, vs] <- $rows\n:put vt_rows {k => vs}",
                    params=dict(rows=[[i, [f32_list(v) for v in vs]] for i, vs in enumerate(d["rows_vectors"])])))
    out.append(dict(name="rows create index", mutable=True,
                    script=f"::hnsw create vt_rows:idx {{dim: {ROWS['dim']}, m: {ROWS['m']}, dtype: F32, fields: [vs], distance: L2, "
                           f"ef_construction: {ROWS['ef_construction']}}}"))
    out.append(dict(name="rows index rows", script=f"?[{all_cols}] := *vt_rows:idx{{{all_cols}}}"))
    out.append(dict(name="rows knn",
                    script=f"qs[qi, qv] <- $queries\n"
                           f"?[qi, k, sub, dist] := qs[qi, qv], q = vec(qv), ~vt_rows:idx{{k | query: q, k: {ROWS['k']}, ef: {ROWS['ef']}, "
                           f"bind_distance: dist, bind_field_idx: sub}}\n"
                           f":order qi, dist, k, sub",
                    params=dict(queries=[[i, f32_list(v)] for i, v in enumerate(d["rows_queries"])])))
    out.append(dict(name="graph create", mutable=True, script=":create edges {fr: Int, to: Int => w: Float}"))
    out.append(dict(name="graph put", mutable=True, script="?[fr, to, w] <- $rows\n:put edges {fr, to => w}",
                    params=dict(rows=[[int(f), int(t), float(w)] for f, t, w in zip(d["frm"], d["to"], d["w"])])))
    out.append(dict(name="pagerank defaults", script="?[n, r] <~ PageRank(*edges[fr, to])"))
    out.append(dict(name="pagerank theta=0.5 4 iterations", script="?[n, r] <~ PageRank(*edges[fr, to], theta: 0.5, epsilon: 0, iterations: 4)"))
    out += pagerank_big_steps(d)
    out.append(dict(name="connected components", script="?[n, g] <~ ConnectedComponents(*edges[fr, to])"))
    out.append(dict(name="dijkstra", script=f"start[] <- [[{d['dijkstra_start']}]]\n?[s, t, c, p] <~ ShortestPathDijkstra(*edges[fr, to, w], start[])"))
    out.append(dict(name="shortest path bfs",
                    script=f"start[] <- [[{d['bfs_start']}]]\ngoal[] <- [{', '.join('[%d]' % g for g in d['bfs_goals'])}]\n"
                           f"?[s, g, p] <~ ShortestPathBFS(*edges[fr, to], start[], goal[])"))
    return out


def pagerank_big_steps(d):
    out = [dict(name="big graph create", mutable=True, script=":create edges_big {fr: Int, to: Int}"),
           dict(name="big graph put", mutable=True, script="?[fr, to] <- $rows\n:put edges_big {fr, to}",
                params=dict(rows=[[int(f), int(t)] for f, t in zip(d["big_frm"], d["big_to"])]))]
    for i in (1, 2, 3):  # the same script three times: a schedule-dependent loop gives three different answers
        out.append(dict(name=f"pagerank big run {i}", script="?[n, r] <~ PageRank(*edges_big[fr, to])"))
    out.append(dict(name="pagerank big converged", script="?[n, r] <~ PageRank(*edges_big[fr, to], epsilon: 0.000000001, iterations: 200)"))
    return out


def steps_pagerank_only():
    """the PageRank steps alone, for the second run of the recipe under RAYON_NUM_THREADS=1"""
    d = dataset()
    out = [dict(name="graph create", mutable=True, script=":create edges {fr: Int, to: Int => w: Float}"),
           dict(name="graph put", mutable=True, script="?[fr, to, w] <- $rows\n:put edges {fr, to => w}",
                params=dict(rows=[[int(f), int(t), float(w)] for f, t, w in zip(d["frm"], d["to"], d["w"])])),
           dict(name="pagerank defaults", script="?[n, r] <~ PageRank(*edges[fr, to])"),
           dict(name="pagerank theta=0.5 4 iterations", script="?[n, r] <~ PageRank(*edges[fr, to], theta: 0.5, epsilon: 0, iterations: 4)")]
    return out + pagerank_big_steps(d)


if __name__ == "__main__":
    only = "--pagerank-only" in sys.argv
    argv = [a for a in sys.argv[1:] if a != "--pagerank-only"]
    path = argv[0] if argv else "inputs.json"
    with open(path, "w") as f:
        json.dump({"steps": steps_pagerank_only() if only else steps()}, f)
    print("wrote", path)

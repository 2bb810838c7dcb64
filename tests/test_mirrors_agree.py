"""The two executable host mirrors of cozo-core's FixedRule surface -- Python (cozo_amd/fixed_rule.py) and C++
(cozo_amd/host -> libcozo_host.so) -- run the SAME rule invocation and must return the SAME rows, byte for byte in the store's
own key encoding (types included: an Int is not a Float).  The invocation travels as stored-key bytes (the codec both sides
already share, tests/cpp/test_host.cpp `run-rule`).  On CPU both mirrors sit on the oracle (tests/util.OracleGraphBackend /
tests/cpp/oracle_shim.c), marked gpu both call libcozo_gpu.so.  A drift in either mirror -- an option default, an id order, a
rounding step, an error code -- fails here."""
import os
import struct
import subprocess

import numpy as np
import pytest

from cozo_amd import codec, fixed_rule as FR
from tests import util
from tests.test_cpp_host import build_test_host, build_test_host_shim

MAGIC = 0x52525A43


def _blob(b: bytes) -> bytes:
    return struct.pack("<I", len(b)) + b


def run_cpp(tmp_path, gpu, name, inputs, options):
    exe = build_test_host() if gpu else build_test_host_shim()
    buf = bytearray(struct.pack("<I", MAGIC)) + _blob(name.encode())
    buf += struct.pack("<I", len(options))
    for k, v in options.items():
        buf += _blob(k.encode()) + _blob(codec.memcmp_bytes(v))
    buf += struct.pack("<I", len(inputs))
    for rows in inputs:
        buf += struct.pack("<I", len(rows))
        for r in rows:
            buf += _blob(codec.encode_key_for_store(1, list(r)))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(bytes(buf))
    if fout.exists():
        fout.unlink()
    p = subprocess.run([exe, "run-rule-gpu" if gpu else "run-rule", str(fin), str(fout)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = fout.read_bytes()
    ok, = struct.unpack_from("<I", out, 0)
    at = 4
    if not ok:
        n, = struct.unpack_from("<I", out, at)
        return ("error", out[at + 4:at + 4 + n].decode())
    n, = struct.unpack_from("<I", out, at)
    at += 4
    rows = []
    for _ in range(n):
        ln, = struct.unpack_from("<I", out, at)
        rows.append(out[at + 4:at + 4 + ln])
        at += 4 + ln
    return ("rows", rows)


def _py_expr(spec):
    """[op, column, constant] -> a predicate over the bound tuple in DataValue's total order (the wire form of an expression
    option, decoded the same way by tests/cpp/test_host.cpp `run-rule`)"""
    op, col, const = spec
    kc = FR.sort_key(const)
    cmp = {"eq": lambda k: k == kc, "ne": lambda k: k != kc, "lt": lambda k: k < kc, "ge": lambda k: k >= kc,
           "gt": lambda k: k > kc, "le": lambda k: k <= kc}[op]

    def pred(t):
        return cmp(FR.sort_key(t[col]))
    pred.only_node_id = col == 0
    return pred


def run_python(registry, name, inputs, options):
    options = {(k[5:] if k.startswith("expr:") else k): (_py_expr(v) if k.startswith("expr:") else v) for k, v in options.items()}
    try:
        rows = registry.run(name + "Gpu", [FR.FixedRuleInputRelation(list(r)) for r in inputs], dict(options))
    except FR.FixedRuleError as e:
        return ("error", e.code)
    return ("rows", [codec.encode_key_for_store(0, list(r)) for r in rows])


def _cases():
    rng = np.random.default_rng(2024)
    ew = [(f"n{int(a)}", f"n{int(b)}", float(rng.integers(1, 9)) / 2) for a, b in rng.integers(0, 40, (160, 2))]
    e = [(a, b) for a, b, _ in ew]
    ints = [(int(a), int(b)) for a, b in rng.integers(0, 60, (220, 2))]
    mixed = [(1, 2.5), (2.5, "x"), ("x", 1), (1, 1), (3, 1)]  # ids of different DataValue types
    starts, goals = [("n1",), ("n7",), ("nowhere",)], [("n3",), ("n9",), ("n1",)]
    cases = [
        ("PageRank", [e], {}),
        ("PageRank", [ints], {"undirected": True, "theta": 0.7, "epsilon": 1e-6, "iterations": 20}),
        ("PageRank", [mixed], {}),
        ("PageRank", [ints], {"in_place": True}),                      # the other reading of graph::page_rank (cz_pagerank_inplace)
        ("PageRank", [e], {"in_place": True, "undirected": True, "iterations": 4, "epsilon": 0.0}),
        ("PageRank", [[]], {}),
        ("ConnectedComponents", [ints], {}),
        ("ConnectedComponents", [e, [("lonely",), ("n3",)]], {}),
        ("ShortestPathBFS", [e, starts, goals], {}),
        ("ShortestPathDijkstra", [ew, starts, goals], {}),
        ("ShortestPathDijkstra", [ew, starts], {"undirected": True}),
        ("ShortestPathDijkstra", [ew, starts, goals], {"keep_ties": True}),
        ("ClusteringCoefficients", [ints], {}),
        ("ClusteringCoefficients", [mixed], {}),
        # Bfs (bfs.rs:25-113): `condition` over the node id alone (no node lookup), over a column of `nodes`, with a limit, with a
        # starting relation of its own, and the NodeNotFoundError of a target that `nodes` does not hold
        ("BFS", [ints, [(int(a),) for a in range(0, 60, 7)]], {"expr:condition": ["gt", 0, 50], "limit": 4}),
        ("BFS", [ints, [(i, i % 5) for i in range(60)], [(3,), (17,), (3,)]], {"expr:condition": ["eq", 1, 2], "limit": 6}),
        ("BFS", [e, [(f"n{i}", i) for i in range(40)], [("n1",), ("n7",)]], {"expr:condition": ["ge", 1, 30]}),
        ("BFS", [ints, [(i, i) for i in range(30)], [(3,)]], {"expr:condition": ["eq", 1, 1000], "limit": 2}),
        ("ShortestPathBFS", [ints, [(5,), (9,)], [(5,), (11,), (58,), (1000,)]], {}),
        ("ShortestPathDijkstra", [ew, starts, goals], {"undirected": True, "keep_ties": True}),
        ("ShortestPathDijkstra", [ew, starts], {"keep_ties": True}),  # without a termination relation keep_ties has no effect (:73-86)
        ("DegreeCentrality", [e], {}),
        ("ClosenessCentrality", [ew], {"undirected": True}),
        ("BetweennessCentrality", [ew], {}),
        ("BetweennessCentrality", [ew], {"undirected": True}),
        ("LabelPropagation", [ew], {"undirected": True, "max_iter": 5}),
        ("LabelPropagation", [ints], {}),
        # diagnostics: the same code out of both mirrors
        ("PageRank", [[("a",)]], {}),                                      # not an edge
        ("ShortestPathDijkstra", [[("a", "b", -1.0)], [("a",)]], {}),      # negative weight
        ("ShortestPathDijkstra", [[("a", "b", "w")], [("a",)]], {}),       # not a number
        ("PageRank", [e], {"iterations": 0}),                              # a positive integer is required
        ("PageRank", [e], {"theta": 1.5}),
        ("LabelPropagation", [ew], {"max_iter": "ten"}),
    ]
    return [pytest.param(*c, id=f"{i:02d}-{c[0]}") for i, c in enumerate(cases)]


@pytest.fixture(params=[pytest.param(False, id="host-logic"), pytest.param(True, marks=pytest.mark.gpu, id="gpu")])
def on_gpu(request, monkeypatch, oracle):
    if request.param:
        request.getfixturevalue("gpu_lib")
    else:
        util.OracleGraphBackend(oracle).install(monkeypatch)
    return request.param


@pytest.mark.parametrize("name,inputs,options", _cases())
def test_python_and_cpp_mirrors_return_the_same_rows(tmp_path, on_gpu, name, inputs, options):
    want = run_python(FR.FixedRuleRegistry(), name, inputs, options)
    got = run_cpp(tmp_path, on_gpu, name, inputs, options)
    assert got[0] == want[0], (got[:1], want[:1], got[1] if got[0] == "error" else "", want[1] if want[0] == "error" else "")
    if want[0] == "error":
        assert got[1] == want[1]
    else:
        assert len(got[1]) == len(want[1])
        for a, b in zip(got[1], want[1]):
            assert a == b, (codec.decode_tuple_from_key(a), codec.decode_tuple_from_key(b))


# ---- the C++ mirror's declared gaps (the Python mirror is the normative one) -----------------------------------
# What the compiled twin does NOT cover, stated where its scope is frozen (cozo_amd/host/include/cozo_host/hnsw.hpp).  Each entry
# must be REFUSED or ABSENT there -- never silently different -- and present in the Python mirror.
CPP_MIRROR_GAPS = {
    "F64 indices": dict(cpp_refuses='only F32 vector indices are GPU-resident', python_has="cz_hnsw_search_batch_f64"),
    "resident in-place PageRank plan": dict(cpp_absent="cz_pagerank_inplace_plan_create", python_has="cz_pagerank_inplace_plan_create"),
    "distance batch on an index table": dict(cpp_absent="cz_hnsw_index_distance_batch", python_has="cz_hnsw_index_distance_batch"),
    "placement by trial": dict(cpp_absent="cz_hnsw_index_settle", python_has="cz_hnsw_index_settle"),
}


def test_cpp_mirror_gaps_are_declared_and_refused():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "cozo_amd", "host")
    cpp = ""
    for d, _, files in os.walk(host):
        for f in files:
            if f.endswith((".cpp", ".hpp")):
                cpp += open(os.path.join(d, f)).read()
    header = open(os.path.join(host, "include", "cozo_host", "hnsw.hpp")).read()
    py = "".join(open(os.path.join(root, "cozo_amd", f)).read() for f in ("hnsw.py", "graph.py", "_lib.py"))
    assert "NORMATIVE executable host mirror is the Python one" in header
    for what, g in CPP_MIRROR_GAPS.items():
        assert g["python_has"] in py, what
        if "cpp_refuses" in g:  # both ways into a C++ GpuHnswIndex (create, from_stored) throw for a non-F32 manifest
            assert cpp.count(g["cpp_refuses"]) >= 2, what
        else:  # named in the scope comment, called nowhere in the C++ mirror
            assert g["cpp_absent"].replace("_create", "") in header or g["cpp_absent"] in header, what
            code = "\n".join(line for line in cpp.split("\n") if not line.lstrip().startswith("//"))
            assert g["cpp_absent"] not in code, what

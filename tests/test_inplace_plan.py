"""The static layout of the level-scheduled Gauss-Seidel PageRank sweep (cozo_amd/csrc/inplace_plan.hpp: levels, level-major
numbering, slices, row blocks, the X / Y / urgent classes of the edges, the two Y streams) checked on the CPU: tests/cpp/
inplace_plan_test.cpp walks the plan's arrays the way the kernels of csrc/pagerank_inplace.hip do, under the earliest and the latest
schedule the device may run, and every score must equal the oracle's restatement of the in-place reading of graph::page_rank
(orc_pagerank_mode(ORC_PR_INPLACE); fixed_rule/algos/pagerank.rs:47-50) bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "inplace_plan_test.cpp")
HDR = os.path.join(ROOT, "cozo_amd", "csrc", "inplace_plan.hpp")
SO = os.path.join(ROOT, "tests", "cpp", "bin", "libinplace_plan_test.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-ffp-contract=off", SRC, "-o", SO])
    return SO


@pytest.fixture(scope="module")
def ipt():
    L = C.CDLL(build())
    L.ipt_emulate.restype = C.c_int
    L.ipt_emulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                              C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def emulate(L, g, tile=4096, rows=256, slice_=16384, part=16384, gap=1, damping=0.85, sweeps=3, schedule=0):
    n = g["n"]
    ioff = np.ascontiguousarray(g["ioff"], dtype=np.uint64)
    isrc = np.ascontiguousarray(g["isrc"], dtype=np.uint32)
    od = np.ascontiguousarray(g["outdeg"], dtype=np.uint32)
    scores = np.empty(n, dtype=np.float32)
    err = C.c_double(0)
    info = np.zeros(8, dtype=np.uint64)
    rc = L.ipt_emulate(ioff.ctypes.data, isrc.ctypes.data, od.ctypes.data, n, tile, rows, slice_, part, gap, damping, sweeps, schedule,
                       scores.ctypes.data, C.byref(err), info.ctypes.data)
    assert rc == 0, rc
    return scores, err.value, dict(zip(("levels", "blocks", "items", "long_rows", "x", "y", "urgent", "long_edges"), (int(v) for v in info)))


def skewed_relation(n, e, seed):
    """hubs at the low ids (like R-MAT without a permutation): rows far longer than a small tile"""
    rng = np.random.default_rng(seed)
    src = (rng.random(e) ** 3 * n).astype(np.int64)
    dst = (rng.random(e) ** 3 * n).astype(np.int64)
    keep = src != dst
    rows = np.unique(np.stack([src[keep], dst[keep]], 1), axis=0)
    return rows[:, 0], rows[:, 1]


@pytest.mark.parametrize("kind,n,e", [("uniform", 3000, 20000), ("uniform", 20000, 200000), ("skewed", 4000, 60000), ("chainy", 600, 3000)])
@pytest.mark.parametrize("gap", [0, 1, 2])
@pytest.mark.parametrize("schedule", [0, 1])
def test_plan_walk_equals_the_oracle(ipt, oracle, kind, n, e, gap, schedule):
    if kind == "uniform":
        frm, to = util.random_relation(n, e, 5)
    elif kind == "skewed":
        frm, to = skewed_relation(n, e, 6)
    else:  # many levels: mostly i -> i + small
        rng = np.random.default_rng(7)
        a = rng.integers(0, n - 8, e)
        frm, to = np.unique(np.stack([a, a + rng.integers(1, 8, e)], 1), axis=0).T
    g = util.graph_from_relation(oracle, frm, to)
    for sweeps in (1, 3):
        want, _, werr = oracle.pagerank_mode(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, 0.0, sweeps, mode=oracle.PR_INPLACE)
        # small tiles / slices / parts so that a small graph has many blocks, slices, items -- and long rows
        got, err, info = emulate(ipt, g, tile=128 if kind != "uniform" else 512, rows=16, slice_=64, part=32, gap=gap, sweeps=sweeps,
                                 schedule=schedule)
        assert np.array_equal(got, want), (info, np.flatnonzero(got != want)[:5])
        assert abs(err - werr) <= 1e-9 * max(1.0, abs(werr))
        assert info["x"] + info["y"] + info["urgent"] + info["long_edges"] == len(g["isrc"])
        if gap == 0:
            assert info["urgent"] == 0
    if kind == "skewed":
        assert info["long_rows"] > 0
    if kind == "chainy":
        assert info["levels"] > 50


def test_default_shape_and_duplicates(ipt, oracle):
    """the default tile / slice / part; parallel edges and a self loop (kept by CsrLayout::Sorted: a self loop reads the OLD value)"""
    frm, to = util.random_relation(5000, 60000, 11, self_loops=True)
    g = util.graph_from_relation(oracle, frm, to)
    # parallel edges: repeat a few in-edges in place (the lists stay ascending)
    ioff, isrc = g["ioff"].astype(np.int64), g["isrc"]
    rep = np.ones(len(isrc), dtype=np.int64)
    rep[::97] = 2
    isrc2 = np.repeat(isrc, rep)
    csum = np.concatenate([[0], np.cumsum(rep)])
    ioff2 = csum[ioff]
    od2 = np.bincount(isrc2, minlength=g["n"]).astype(np.uint32)
    g2 = dict(n=g["n"], ioff=ioff2.astype(np.uint64), isrc=isrc2, outdeg=od2)
    want, _, _ = oracle.pagerank_mode(g2["n"], g2["ioff"], g2["isrc"], g2["outdeg"], 0.85, 0.0, 4, mode=oracle.PR_INPLACE)
    for schedule in (0, 1):
        got, _, info = emulate(ipt, g2, sweeps=4, schedule=schedule)
        assert np.array_equal(got, want)


def test_empty_and_edgeless(ipt, oracle):
    g = dict(n=0, ioff=np.zeros(1, np.uint64), isrc=np.zeros(0, np.uint32), outdeg=np.zeros(0, np.uint32))
    got, _, info = emulate(ipt, g)
    assert got.size == 0 and info["levels"] == 0
    g = dict(n=5, ioff=np.zeros(6, np.uint64), isrc=np.zeros(0, np.uint32), outdeg=np.zeros(5, np.uint32))
    want, _, _ = oracle.pagerank_mode(5, g["ioff"], g["isrc"], g["outdeg"], 0.85, 0.0, 2, mode=oracle.PR_INPLACE)
    got, _, info = emulate(ipt, g, sweeps=2)
    assert np.array_equal(got, want) and info["levels"] == 1


def test_jacobi_reading_in_the_same_layout(ipt, oracle):
    """prm.jacobi: one level, every edge reads the previous sweep's contribution -- the walk must equal orc_pagerank (the Jacobi
    reading), the layout being the grouped tile formulation csrc/pagerank_inplace.hip runs"""
    for kind, (n, e) in (("uniform", (20000, 200000)), ("skewed", (4000, 60000))):
        frm, to = util.random_relation(n, e, 5) if kind == "uniform" else skewed_relation(n, e, 6)
        g = util.graph_from_relation(oracle, frm, to)
        want, _, _ = oracle.pagerank(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, 0.0, 3)
        got, _, info = emulate(ipt, g, tile=512, rows=16, slice_=64, part=32, gap=1 | (1 << 16), sweeps=3)
        assert info["levels"] == 1 and info["x"] == 0 and info["urgent"] == 0
        assert np.array_equal(got, want)


def test_layout_is_the_same_for_every_thread_count(ipt, oracle):
    """the two edge passes of build_plan run on several host threads over contiguous block ranges; a graph large enough to take the
    threaded path must give the oracle's scores too (the small graphs above all run on one thread)"""
    frm, to = util.random_relation(150000, 1400000, 21)
    g = util.graph_from_relation(oracle, frm, to)
    assert len(g["isrc"]) >= (1 << 20)
    want, _, _ = oracle.pagerank_mode(g["n"], g["ioff"], g["isrc"], g["outdeg"], 0.85, 0.0, 2, mode=oracle.PR_INPLACE)
    got, _, info = emulate(ipt, g, tile=2048, rows=256, slice_=4096, part=2048, gap=1, sweeps=2)
    assert info["blocks"] > 64 and np.array_equal(got, want)

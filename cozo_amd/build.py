"""Builds cozo_amd/lib/libcozo_gpu.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m cozo_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the arithmetic contract: the kernels
spell every fma explicitly (distance.h) and PageRank's `base + d*sum` must round twice like the reference.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
SO = os.path.join(LIBDIR, "libcozo_gpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip")) and f != "cozo_ingest.h":
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= _deps_mtime():
        return obj
    subprocess.check_call([HIPCC, *FLAGS, "-c", src, "-o", obj])
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= _deps_mtime():
        return SO
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    # Linked WITHOUT a NEEDED entry for libamdhip64: a process must hold exactly one HIP/HSA runtime, and
    # PyTorch wheels bundle their own (torch/lib/libamdhip64.so).  The loader (cozo_amd/_lib.py, or the host
    # binary that links -lamdhip64 itself) brings the runtime in first; see INTEGRATION.md.
    subprocess.check_call([os.environ.get("CXX", "g++"), "-shared", "-fPIC", *objs, "-o", SO])
    if verbose:
        print("built", SO)
    return SO




# ---- C++ host mirror (cozo_amd/host): libcozo_host.so, the compiled host side above the C ABI ------------------
HOST = os.path.join(HERE, "host")
HOST_SO = os.path.join(LIBDIR, "libcozo_host.so")
CXX = os.environ.get("CXX", "g++")
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter",
              "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(HERE, "..", "include")]


def _host_mtime():
    m = max(os.path.getmtime(os.path.join(HERE, "..", "include", "cozo_gpu.h")),
            os.path.getmtime(os.path.join(HERE, "..", "include", "cozo_ingest.h")))
    for root, _, files in os.walk(HOST):
        for f in files:
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build_host(force: bool = False, verbose: bool = False) -> str:
    """g++ only (no device code).  Links against libcozo_gpu.so ($ORIGIN rpath); the HIP runtime is left to the
    final executable / the process, exactly like libcozo_gpu.so itself."""
    so = build(force=False)
    ingest_so = build_ingest(force=False)
    if not force and os.path.exists(HOST_SO) and os.path.getmtime(HOST_SO) >= max(_host_mtime(), os.path.getmtime(so),
                                                                                os.path.getmtime(ingest_so)):
        return HOST_SO
    srcs = sorted(os.path.join(HOST, "src", f) for f in os.listdir(os.path.join(HOST, "src")) if f.endswith(".cpp"))
    subprocess.check_call([CXX, *HOST_FLAGS, "-shared", *srcs, "-o", HOST_SO, "-L" + LIBDIR, "-lcozo_gpu", "-lcozo_ingest",
                           "-Wl,-rpath,$ORIGIN", "-Wl,--allow-shlib-undefined"])
    if verbose:
        print("built", HOST_SO)
    return HOST_SO


# ---- host ingest (cozo_amd/ingest): libcozo_ingest.so, stored rows -> flat arrays; no device code, no HIP ----------
INGEST_DIR = os.path.join(HERE, "ingest")
INGEST_SO = os.path.join(LIBDIR, "libcozo_ingest.so")


def ingest_sources():
    return sorted(os.path.join(INGEST_DIR, f) for f in os.listdir(INGEST_DIR) if f.endswith(".cpp"))


def build_ingest(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    inc = os.path.join(HERE, "..", "include")
    deps = [os.path.join(INGEST_DIR, f) for f in os.listdir(INGEST_DIR) if f.endswith((".cpp", ".hpp"))]
    deps += [os.path.join(inc, "cozo_ingest.h"), os.path.join(inc, "cozo_gpu.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    if not force and os.path.exists(INGEST_SO) and os.path.getmtime(INGEST_SO) >= newest:
        return INGEST_SO
    subprocess.check_call([CXX, "-std=c++17", "-O2", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra", "-pthread", "-shared",
                           "-I" + inc, *ingest_sources(), "-o", INGEST_SO])
    if verbose:
        print("built", INGEST_SO)
    return INGEST_SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_host(force="--force" in sys.argv, verbose=True)
    build_ingest(force="--force" in sys.argv, verbose=True)

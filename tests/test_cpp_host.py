"""C++ host mirror (cozo_amd/host -> libcozo_host.so) driven by tests/cpp/test_host.cpp.

The binary is built in-tree (tests/cpp/bin/test_host, git-ignored, travels to the GPU box with the snapshot) from
the host library, the C ABI library and the CPU oracle; `cpu` mode needs no device, `gpu` mode runs the fixed
rules and HnswSearchRA on the MI355X and compares their rows with the oracle's."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin", "test_host")
SRC = os.path.join(ROOT, "tests", "cpp", "test_host.cpp")
ROCM_LIB = os.environ.get("ROCM_PATH", "/opt/rocm") + "/lib"


def build_test_host():
    from cozo_amd import build as B
    from oracle import oracle as O
    host_so = B.build_host()
    O.build()
    libdir = os.path.dirname(host_so)
    ordir = os.path.join(ROOT, "oracle")
    deps = [SRC, host_so, os.path.join(libdir, "libcozo_gpu.so"), os.path.join(ordir, "libcozo_oracle.so")]
    if os.path.exists(BIN) and os.path.getmtime(BIN) >= max(os.path.getmtime(d) for d in deps):
        return BIN
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    # the executable is what brings the HIP runtime into the process (libcozo_gpu.so has no NEEDED entry for it)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "cozo_amd", "host", "include"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + ordir, SRC, "-o", BIN,
                           "-L" + libdir, "-lcozo_host", "-lcozo_gpu", "-lcozo_ingest", "-L" + ordir, "-lcozo_oracle",
                           "-L" + ROCM_LIB, "-lamdhip64", "-Wl,--no-as-needed",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + ordir, "-Wl,-rpath," + ROCM_LIB])
    return BIN


SHIM_BIN = os.path.join(ROOT, "tests", "cpp", "bin", "test_host_shim")


def build_test_host_shim():
    """test_host linked against the oracle shim AHEAD of libcozo_gpu.so (symbol interposition): the graph entry points
    resolve to the CPU oracle, so the C++ rules' host logic runs without a device.  TEST ONLY."""
    build_test_host()
    from cozo_amd import build as B
    libdir = os.path.dirname(B.build_host())
    ordir = os.path.join(ROOT, "oracle")
    shim_src = os.path.join(ROOT, "tests", "cpp", "oracle_shim.c")
    shim_so = os.path.join(ROOT, "tests", "cpp", "bin", "libcozo_gpu_shim.so")
    deps = [SRC, shim_src, os.path.join(libdir, "libcozo_host.so"), os.path.join(ordir, "libcozo_oracle.so")]
    if os.path.exists(SHIM_BIN) and os.path.getmtime(SHIM_BIN) >= max(os.path.getmtime(d) for d in deps):
        return SHIM_BIN
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + ordir, shim_src,
                           "-o", shim_so, "-L" + ordir, "-lcozo_oracle", "-Wl,-rpath," + ordir])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "cozo_amd", "host", "include"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + ordir, SRC, "-o", SHIM_BIN,
                           "-Wl,--no-as-needed", "-L" + os.path.dirname(shim_so), "-lcozo_gpu_shim", "-L" + libdir, "-lcozo_host", "-lcozo_gpu", "-lcozo_ingest",
                           "-L" + ordir, "-lcozo_oracle", "-L" + ROCM_LIB, "-lamdhip64",
                           "-Wl,-rpath," + os.path.dirname(shim_so), "-Wl,-rpath," + libdir, "-Wl,-rpath," + ordir,
                           "-Wl,-rpath," + ROCM_LIB])
    return SHIM_BIN


def run(mode):
    exe = build_test_host_shim() if mode == "rules-cpu" else build_test_host()
    p = subprocess.run([exe, mode], capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "0 failed" in p.stdout


def test_cpp_host_logic_cpu():
    run("cpu")


def test_cpp_rules_host_logic_with_oracle_shim():
    """the C++ rules (PageRank, ShortestPathBFS, BFS, ConnectedComponents, Dijkstra, ClusteringCoefficients,
    ClosenessCentrality) end to end on CPU: the device entry points are interposed by tests/cpp/oracle_shim.c"""
    run("rules-cpu")


@pytest.mark.gpu
def test_cpp_host_rules_gpu():
    run("gpu")


def test_cpp_host_under_sanitizers():
    """the whole compiled host side -- rules, codec, libcozo_ingest's sources, with the oracle shim standing in for the device
    entry points -- built with -fsanitize=address,undefined and run through the rules-cpu checks: no finding"""
    from cozo_amd import build as B
    B.build()
    libdir = os.path.join(ROOT, "cozo_amd", "lib")
    ordir = os.path.join(ROOT, "oracle")
    bindir = os.path.join(ROOT, "tests", "cpp", "bin")
    os.makedirs(bindir, exist_ok=True)
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    hostdir = os.path.join(ROOT, "cozo_amd", "host")
    host_src = sorted(os.path.join(hostdir, "src", f) for f in os.listdir(os.path.join(hostdir, "src")) if f.endswith(".cpp"))
    c_src = [(os.path.join(ROOT, "tests", "cpp", "oracle_shim.c"), ["-I" + os.path.join(ROOT, "include"), "-I" + ordir]),
             (os.path.join(ordir, "cozo_oracle.c"), ["-I" + ordir, "-fopenmp"])]
    deps = [SRC, *host_src, *B.ingest_sources(), *(c for c, _ in c_src), os.path.join(B.INGEST_DIR, "common.hpp"),
            os.path.join(ordir, "cozo_oracle.h"), os.path.join(ROOT, "include", "cozo_gpu.h"), os.path.join(ROOT, "include", "cozo_ingest.h")]
    deps += [os.path.join(hostdir, "include", "cozo_host", f) for f in os.listdir(os.path.join(hostdir, "include", "cozo_host"))]
    exe = os.path.join(bindir, "test_host_san")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        objs = []
        for src, extra in c_src:
            obj = os.path.join(bindir, os.path.basename(src) + ".san.o")
            subprocess.check_call(["gcc", "-c", *san, *extra, src, "-o", obj])
            objs.append(obj)
        subprocess.check_call(["g++", "-std=c++17", *san, "-pthread", "-fopenmp", "-I" + os.path.join(hostdir, "include"),
                               "-I" + os.path.join(ROOT, "include"), "-I" + ordir, SRC, *host_src, *B.ingest_sources(), *objs, "-o", exe,
                               "-L" + libdir, "-lcozo_gpu", "-L" + ROCM_LIB, "-lamdhip64", "-lm", "-Wl,-rpath," + libdir,
                               "-Wl,-rpath," + ROCM_LIB])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", LD_PRELOAD="")
    p = subprocess.run([exe, "rules-cpu"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "0 failed" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]

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
                           "-L" + libdir, "-lcozo_host", "-lcozo_gpu", "-L" + ordir, "-lcozo_oracle",
                           "-L" + ROCM_LIB, "-lamdhip64", "-Wl,--no-as-needed",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + ordir, "-Wl,-rpath," + ROCM_LIB])
    return BIN


def run(mode):
    exe = build_test_host()
    p = subprocess.run([exe, mode], capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "0 failed" in p.stdout


def test_cpp_host_logic_cpu():
    run("cpu")


@pytest.mark.gpu
def test_cpp_host_rules_gpu():
    run("gpu")

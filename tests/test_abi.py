"""CPU tests of the drop-in boundary: libcozo_gpu.so loads and exports every symbol include/cozo_gpu.h
declares; without a GPU the compute entry points refuse loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cozo_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cozo_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cozo_gpu.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes signature in cozo_amd/_lib.py"
    assert sorted(_lib.SYMBOLS) == declared


def test_ingest_library_exports_every_declared_symbol():
    """include/cozo_ingest.h <-> libcozo_ingest.so <-> cozo_amd/ingest.py (host-only library: no HIP needed to load it)"""
    from cozo_amd import build as B, ingest
    B.build_ingest()
    text = open(os.path.join(ROOT, "include", "cozo_ingest.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(czi_[a-z0-9_]+)\s*\(", text)))
    L = ingest.lib()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cozo_ingest.h but not exported"
    assert sorted(ingest.SYMBOLS) == declared
    assert b"cozo_ingest" in L.czi_version()
    import subprocess
    needed = subprocess.run(["readelf", "-d", ingest.SO_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "cozo_gpu" not in needed  # stands alone


def test_public_headers_are_plain_c(tmp_path):
    """the boundary is a C ABI: both headers compile as strict C99 (no C++-isms, no torch / HIP types in a signature) and a
    C program links against libcozo_ingest.so"""
    import subprocess
    from cozo_amd import build as B, ingest
    B.build_ingest()
    src = tmp_path / "hdr.c"
    src.write_text('#include "cozo_gpu.h"\n#include "cozo_ingest.h"\n#include <stdio.h>\n'
                   'int main(void) { czi_rows r; cz_hnsw_desc d; (void)r; (void)d; puts(czi_version()); return 0; }\n')
    exe = tmp_path / "hdr"
    libdir = os.path.dirname(ingest.SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L" + libdir, "-lcozo_ingest", "-Wl,-rpath," + libdir])
    assert "cozo_ingest" in subprocess.run([str(exe)], capture_output=True, text=True).stdout
    for h in ("cozo_gpu.h", "cozo_ingest.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        assert "torch" not in text and "hipStream_t " not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cozo_amd import _lib, graph as G
    from cozo_amd.hnsw import distance_batch
    L = _lib.lib()
    assert L.cz_device_count() == 0
    assert L.cz_init(0) == _lib.CZ_E_NO_DEVICE
    assert b"no CPU fallback" in L.cz_last_error()
    with pytest.raises(_lib.CozoGpuError) as ei:
        distance_batch("L2", np.ones((2, 4), np.float32), np.ones((1, 4), np.float32), np.zeros((1, 2), np.uint32))
    assert ei.value.code == _lib.CZ_E_NO_DEVICE
    with pytest.raises(_lib.CozoGpuError):
        G.pagerank(np.array([0, 1, 2], np.uint32), np.array([1, 0], np.uint32), np.array([1, 1], np.uint32))


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under cozo_amd/ (or the C ABI sources) may reference it."""
    pkg = os.path.join(ROOT, "cozo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                bad = re.search(r"^\s*(from\s+oracle|import\s+oracle)|#\s*include\s*[<\"].*oracle|dlopen.*oracle|"
                                r"CDLL\(.*oracle", text, flags=re.M)
                assert bad is None, (os.path.join(dirpath, f), bad.group(0))


def test_graft_entry_build_is_idempotent():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(os.path.join(ROOT, "cozo_amd", "lib", "libcozo_gpu.so"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "libcozo_oracle.so"))
    assert os.path.exists(os.path.join(ROOT, "cozo_amd", "lib", "libcozo_ingest.so"))

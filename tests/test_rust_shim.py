"""integration/rust/cozo_gpu_sys.rs (the reference-side FFI declarations; not compilable here -- no rustc) stays in step
with the C headers: every declared symbol, with the same number of parameters, and the #[repr(C)] structs field by field."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _c_functions(header, prefix):
    text = _strip_c(open(os.path.join(ROOT, "include", header)).read())
    out = {}
    for m in re.finditer(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_functions():
    text = re.sub(r"//.*", "", open(os.path.join(ROOT, "integration", "rust", "cozo_gpu_sys.rs")).read())
    out = {}
    for m in re.finditer(r"pub fn ([a-z0-9_]+)\s*\(([^)]*)\)", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out, text


def test_every_c_symbol_is_declared_with_the_same_arity():
    rust, _ = _rust_functions()
    c = {}
    c.update(_c_functions("cozo_gpu.h", "cz_"))
    c.update(_c_functions("cozo_ingest.h", "czi_"))
    assert len(c) >= 45
    assert sorted(rust) == sorted(c)
    for name, n in c.items():
        assert rust[name] == n, (name, rust[name], n)


def test_repr_c_structs_match_field_by_field():
    _, rust = _rust_functions()
    for header, struct in (("cozo_gpu.h", "cz_hnsw_desc"), ("cozo_ingest.h", "czi_rows")):
        text = _strip_c(open(os.path.join(ROOT, "include", header)).read())
        body = re.search(r"typedef struct \{(.*?)\}\s*" + struct + r"\s*;", text, flags=re.S).group(1)
        c_fields = [re.sub(r".*[\s\*]", "", f.strip()) for f in body.split(";") if f.strip()]
        rbody = re.search(r"pub struct " + struct + r"\s*\{(.*?)\}", rust, flags=re.S).group(1)
        r_fields = [f.split(":")[0].replace("pub", "").strip() for f in rbody.split(",") if ":" in f]
        assert c_fields == r_fields, (struct, c_fields, r_fields)


def test_constants_match():
    _, rust = _rust_functions()
    for header in ("cozo_gpu.h", "cozo_ingest.h"):
        text = _strip_c(open(os.path.join(ROOT, "include", header)).read())
        for name, val in re.findall(r"#define\s+(CZI?_[A-Z_]+)\s+(0x[0-9A-Fa-f]+|\d+)u?\b", text):
            m = re.search(r"pub const " + name + r": u32 = ([0-9A-Fa-fx_]+);", rust)
            assert m, name
            assert int(m.group(1).replace("_", ""), 0) == int(val, 0), name
        for name, val in re.findall(r"\b(CZI?_(?:OK|E_[A-Z_]+|L2|COSINE|IP))\s*=\s*(-?\d+)", text):
            m = re.search(r"pub const " + name + r": c_int = (-?\d+);", rust)
            assert m and int(m.group(1)) == int(val), name

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
    for header, struct in (("cozo_gpu.h", "cz_hnsw_desc"), ("cozo_gpu.h", "cz_pagerank_timing"), ("cozo_gpu.h", "cz_predicate"),
                           ("cozo_ingest.h", "czi_rows")):
        text = _strip_c(open(os.path.join(ROOT, "include", header)).read())
        body = re.search(r"typedef struct \{([^}]*)\}\s*" + struct + r"\s*;", text, flags=re.S).group(1)
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


# ---- type by type: every parameter and the return type of every function ------------------------------------------------
_C2R = {"void": "c_void", "int": "c_int", "float": "c_float", "double": "c_double", "char": "c_char", "uint8_t": "u8",
        "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64", "size_t": "usize"}


def _c_type_to_rust(t):
    """`const uint32_t *const *` -> `*const *const u32`; `cz_comm **` -> `*mut *mut cz_comm`; `const volatile uint8_t *` ->
    `*const u8` (volatile is not part of a Rust pointer type)."""
    t = re.sub(r"/\*.*?\*/", "", t).strip()
    t = re.sub(r"\bvolatile\b", "", t)
    t = re.sub(r"\[[^\]]*\]", "*", t)  # array parameter = pointer
    toks = re.findall(r"\*|\w+", t)
    # split at the stars: base part, then one qualifier set per star (the `const` AFTER a star qualifies that pointer)
    parts, cur = [], []
    for tok in toks:
        if tok == "*":
            parts.append(cur)
            cur = []
        else:
            cur.append(tok)
    trailing = cur  # qualifiers after the last star (or the whole type when there is no star)
    if not parts:
        base = [x for x in trailing if x not in ("const", "struct", "enum")]
        assert len(base) == 1, t
        return _C2R.get(base[0], base[0])
    base_toks = parts[0]
    base_const = "const" in base_toks
    base = [x for x in base_toks if x not in ("const", "struct", "enum")]
    assert len(base) == 1, t
    out = _C2R.get(base[0], base[0])
    # pointer levels, innermost first: level i points at something that is const iff the qualifiers BEFORE its star say so
    consts = [base_const] + [("const" in p) for p in parts[1:]]
    for c in consts:
        out = ("*const " if c else "*mut ") + out
    return out


def _c_signatures(header, prefix):
    text = _strip_c(open(os.path.join(ROOT, "include", header)).read())
    out = {}
    for m in re.finditer(r"([\w \*]+?)\b(" + prefix + r"[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret = m.group(1).strip()
        if "typedef" in ret:
            continue
        args = m.group(3).strip()
        params = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = " ".join(a.split())
                # drop the parameter name (the last identifier, unless the declaration is a bare type)
                mm = re.match(r"^(.*?[\*\s])(\w+)(\[[^\]]*\])?$", a)
                ty = (mm.group(1) + (mm.group(3) or "")) if mm else a
                params.append(_c_type_to_rust(ty))
        out[m.group(2)] = (None if ret == "void" else _c_type_to_rust(ret), params)
    return out


def _rust_signatures():
    text = re.sub(r"//.*", "", open(os.path.join(ROOT, "integration", "rust", "cozo_gpu_sys.rs")).read())
    out = {}
    for m in re.finditer(r"pub fn ([a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        params = []
        for a in m.group(2).split(","):
            if a.strip():
                params.append(" ".join(a.split(":", 1)[1].split()))
        ret = " ".join(m.group(3).split()) if m.group(3) else None
        out[m.group(1)] = (ret, params)
    return out


def test_every_parameter_and_return_type_matches():
    """the uncompiled Rust declarations cannot drift from the headers unnoticed: each function's return type and each
    parameter's type, translated from C (uint32_t -> u32, `const T *` -> *const T, `T **` -> *mut *mut T, ...), must be
    literally what cozo_gpu_sys.rs declares"""
    c = {}
    c.update(_c_signatures("cozo_gpu.h", "cz_"))
    c.update(_c_signatures("cozo_ingest.h", "czi_"))
    rust = _rust_signatures()
    assert sorted(c) == sorted(rust)
    bad = []
    for name, (ret, params) in sorted(c.items()):
        rret, rparams = rust[name]
        if ret != rret:
            bad.append((name, "return", ret, rret))
        for i, (a, b) in enumerate(zip(params, rparams)):
            if a != b:
                bad.append((name, i, a, b))
    assert not bad, bad


# ---- the reference-side identifiers the shim files lean on -------------------------------------------------------------
REF = "/root/reference/cozo-core/src"
SHIM_DIR = os.path.join(ROOT, "integration", "rust")


def _shim_sources():
    return {f: re.sub(r"//.*", "", open(os.path.join(SHIM_DIR, f)).read()) for f in sorted(os.listdir(SHIM_DIR)) if f.endswith(".rs")}


def _ref_text():
    out = []
    for d, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".rs"):
                out.append(open(os.path.join(d, f), errors="replace").read())
    return "\n".join(out)


def _module_text(path):
    """source of crate::<path> (a file or a directory module) in the reference, None when there is no such module"""
    base = os.path.join(REF, *path)
    for cand in (base + ".rs", os.path.join(base, "mod.rs")):
        if os.path.exists(cand):
            if cand.endswith("mod.rs"):
                return "\n".join(open(os.path.join(d, f), errors="replace").read() for d, _, fs in os.walk(base) for f in fs if f.endswith(".rs"))
            return open(cand, errors="replace").read()
    return None


import pytest  # noqa: E402


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_shim_only_names_things_the_reference_or_the_patch_defines():
    """Round 2's shim called `store_tx.snapshot_version()`, which `StoreTx` (storage/mod.rs:31-164) does not have, and nothing
    noticed: there is no rustc here.  This is the check that can be made without one: every `crate::` item the shim files
    import or name, and every method they call on the transaction / payload / store objects, must be defined in
    /root/reference/cozo-core/src -- or in integration/rust/db_patch.rs, the file that lists what the patch ADDS."""
    shims = _shim_sources()
    ref_all = _ref_text()
    own = "\n".join(shims.values())
    missing = []
    # 1. `use crate::a::b::{X, Y}` / `use crate::a::b::X` / inline `crate::a::b::X`
    names = set()
    for text in shims.values():
        for m in re.finditer(r"crate::((?:[a-z_0-9]+::)+)\{([^}]*)\}", text):
            for item in m.group(2).split(","):
                item = item.strip().split(" as ")[0].strip()
                if item and item != "self":
                    names.add((tuple(m.group(1).strip(":").split("::")), item))
        for m in re.finditer(r"crate::((?:[a-z_0-9]+::)+)([A-Za-z_][A-Za-z0-9_]*)", text):
            names.add((tuple(m.group(1).strip(":").split("::")), m.group(2)))
    assert len(names) >= 15
    for path, item in sorted(names):
        mod = _module_text(path)
        if mod is None and len(path) > 1:  # crate::a::b::Type::Variant: the last path element is a type, not a module
            mod = _module_text(path[:-1])
        defined = mod is not None and re.search(r"\b(struct|enum|trait|fn|type|const|static|mod|macro_rules!)\s+" + re.escape(item) + r"\b", mod)
        variant = mod is not None and re.search(r"\b" + re.escape(item) + r"\s*[\{\(,]", mod)  # an enum variant / re-export
        in_patch = re.search(r"\b(struct|enum|trait|fn|type|const)\s+" + re.escape(item) + r"\b", own)
        if not (defined or variant or in_patch):
            missing.append("crate::" + "::".join(path) + "::" + item)
    # 2. methods called on the reference's objects
    receivers = r"(?:self\.tx|tx|payload|edges|nodes|self\.tx\.store_tx|self\.tx\.temp_store_tx|store_tx|temp_store_tx|rel|out|poison|self)"
    calls = set()
    for text in shims.values():
        calls.update(re.findall(receivers + r"\.([a-z_][a-z0-9_]*)\s*\(", text))
    assert "get_relation" in calls and "range_scan" in calls
    for name in sorted(calls):
        pat = r"\bfn\s+" + re.escape(name) + r"\b"
        if not (re.search(pat, ref_all) or re.search(pat, own)):
            missing.append("." + name + "()")
    assert not missing, missing
    # and the one that started it stays gone
    assert "snapshot_version" not in own.replace("Round 2's shim called a `snapshot_version()`", "")


# ---- every rule the integration guide lists has its `impl FixedRule`, with the reference's arity and option names -------------
RULES = {  # GPU struct -> (reference file under fixed_rule/algos, reference struct, options the GPU rule may ADD)
    "PageRankGpu": ("pagerank.rs", "PageRank", {"gpus", "in_place"}),
    "ConnectedComponentsGpu": ("strongly_connected_components.rs", "StronglyConnectedComponent", set()),
    "ShortestPathBFSGpu": ("shortest_path_bfs.rs", "ShortestPathBFS", set()),
    "BfsGpu": ("bfs.rs", "Bfs", set()),
    "ShortestPathDijkstraGpu": ("shortest_path_dijkstra.rs", "ShortestPathDijkstra", set()),
    "ClusteringCoefficientsGpu": ("triangles.rs", "ClusteringCoefficients", set()),
    "ClosenessCentralityGpu": ("all_pairs_shortest_path.rs", "ClosenessCentrality", set()),
    "BetweennessCentralityGpu": ("all_pairs_shortest_path.rs", "BetweennessCentrality", set()),
    "LabelPropagationGpu": ("label_propagation.rs", "LabelPropagation", set()),
}


def _impl_block(text, struct):
    m = re.search(r"impl FixedRule for " + struct + r"\s*\{", text)
    assert m, f"no `impl FixedRule for {struct}`"
    depth, i = 1, m.end()
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_every_listed_rule_has_an_impl_with_the_reference_arity_and_options():
    shim = re.sub(r"//.*", "", open(os.path.join(SHIM_DIR, "fixed_rules_gpu.rs")).read())
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for gpu, (ref_file, ref_struct, extra) in RULES.items():
        assert gpu in guide or gpu.replace("Bfs", "BFS") in guide, gpu
        mine = _impl_block(shim, gpu)
        ref = _impl_block(re.sub(r"//.*", "", open(os.path.join(REF, "fixed_rule", "algos", ref_file)).read()), ref_struct)
        opt = lambda t: set(re.findall(r"_option\(\s*\"([a-z_]+)\"", t))
        assert opt(mine) - extra == opt(ref), (gpu, opt(mine), opt(ref))
        arity = lambda t: re.search(r"fn arity\b.*?Ok\((\d+)\)", t, flags=re.S).group(1)
        assert arity(mine) == arity(ref), (gpu, arity(mine), arity(ref))
        # inputs are read by the same positions
        inputs = lambda t: sorted(set(re.findall(r"get_input\((\d)\)", t)))
        assert inputs(mine) == inputs(ref), (gpu, inputs(mine), inputs(ref))

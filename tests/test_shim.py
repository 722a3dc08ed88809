"""include/rayca_shim.rs (the Rust binding a maintainer adds; there is no Rust toolchain here to compile it) against
include/rayca_hip.h: every #[repr(C)] struct must list the C struct's fields in the same order with types of the same size
and alignment, every constant must carry the header's value, and the extern "C" block must declare exactly the header's
entry points with the same number of parameters."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rayca_hip.h")).read()
SHIM = open(os.path.join(ROOT, "include", "rayca_shim.rs")).read()

C_SCALARS = {"uint32_t": ("u32", 4), "int32_t": ("i32", 4), "uint64_t": ("u64", 8), "float": ("f32", 4), "uint8_t": ("u8", 1)}


def strip_c_comments(t):
    return re.sub(r"/\*.*?\*/", "", t, flags=re.S)


def c_structs():
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", strip_c_comments(HEADER), flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(const )?(\w+)\s*(\*)?\s*(\w+)(\[(\d+)\])?$", decl)
            assert fm, decl
            const, ctype, ptr, name, _, n = fm.groups()
            fields.append((name, ctype, bool(ptr), bool(const), int(n) if n else 0))
        out[m.group(1)] = fields
    return out


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", SHIM, flags=re.S):
        fields = []
        for line in m.group(2).splitlines():
            line = line.split("//")[0].strip().rstrip(",")
            if not line:
                continue
            fm = re.match(r"(?:pub )?(\w+): (.+)$", line)
            assert fm, line
            fields.append((fm.group(1), fm.group(2).strip()))
        out[m.group(1)] = fields
    return out


def expected_rust_type(ctype, ptr, const, n, c_names):
    if ptr:
        inner = "c_void" if ctype == "void" else (C_SCALARS[ctype][0] if ctype in C_SCALARS else ctype)
        return f"*{'const' if const else 'mut'} {inner}"
    base = C_SCALARS[ctype][0] if ctype in C_SCALARS else ctype
    assert ctype in C_SCALARS or ctype in c_names, ctype
    return f"[{base}; {n}]" if n else base


def test_every_struct_has_the_headers_fields_in_order():
    cs, rs = c_structs(), rust_structs()
    assert set(cs) <= set(rs), sorted(set(cs) - set(rs))
    assert len(cs) == 17                                   # every typedef struct of the header (RaycaScene is opaque)
    for name, fields in cs.items():
        rf = rs[name]
        assert [f[0] for f in fields] == [f[0] for f in rf], name
        for (fname, ctype, ptr, const, n), (_, rtype) in zip(fields, rf):
            assert rtype == expected_rust_type(ctype, ptr, const, n, cs), f"{name}.{fname}: {rtype}"
    assert rs["RaycaScene"] == [("_private", "[u8; 0]")]


def test_struct_sizes_follow_from_the_same_layout_rule():
    """repr(C) lays a struct out by the C rule; computing that rule over the shim's field list must give the sizes the C
    compiler gives the header (test_abi.py checks those against ctypes)."""
    from rayca_amd import abi
    import ctypes as C
    rs = rust_structs()
    prim = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "u64": (8, 8)}

    def layout(t):
        if t.startswith("*"):
            return 8, 8
        m = re.match(r"\[(\w+); (\d+)\]", t)
        if m:
            s, a = layout(m.group(1))
            return s * int(m.group(2)), a
        if t in prim:
            return prim[t]
        return struct_layout(t)

    def struct_layout(name):
        off, align = 0, 1
        for _, t in rs[name]:
            s, a = layout(t)
            off = (off + a - 1) // a * a + s
            align = max(align, a)
        return (off + align - 1) // align * align, align

    for name in c_structs():
        assert struct_layout(name)[0] == C.sizeof(getattr(abi, name)), name


def test_constants_carry_the_headers_values():
    enums = dict(re.findall(r"\b(RAYCA_[A-Z0-9_]+)\s*=\s*(-?\d+)", strip_c_comments(HEADER)))
    defines = dict(re.findall(r"#define (RAYCA_[A-Z_]+) (0x[0-9A-Fa-f]+|\d+)u", HEADER))
    consts = dict(re.findall(r"pub const (RAYCA_[A-Z0-9_]+): [ui]32 = (-?[0-9xA-Fa-f_]+);", SHIM))
    assert len(consts) > 35
    for name, value in consts.items():
        want = enums.get(name, defines.get(name))
        assert want is not None, name
        assert int(value.replace("_", ""), 0) == int(want, 0), name
    # the enums that cross as `as u32` of the reference's own #[repr(u32)] types need no constant; all others are there
    must = [n for n in enums if not n.startswith(("RAYCA_INTEGRATOR_", "RAYCA_SAMPLER_", "RAYCA_INDEX_"))]
    assert sorted(set(must) - set(consts)) == []


def test_extern_block_declares_exactly_the_entry_points():
    c_fns = {}
    for m in re.finditer(r"\b(?:int32_t|uint32_t|void)\s+(rayca_hip_\w+)\s*\((.*?)\);", strip_c_comments(HEADER), flags=re.S):
        args = m.group(2).strip()
        c_fns[m.group(1)] = 0 if args in ("void", "") else len(args.split(","))
    block = re.search(r'extern "C" \{(.*?)\n\}', SHIM, flags=re.S).group(1)
    r_fns = {}
    for m in re.finditer(r"pub fn (rayca_hip_\w+)\((.*?)\)", block, flags=re.S):
        args = m.group(2).strip()
        r_fns[m.group(1)] = 0 if not args else len(args.split(","))
    assert r_fns == c_fns


def test_flat_scene_fills_every_descriptor_field():
    """FlatScene::desc() must initialise each field of RaycaSceneDesc (a Rust struct literal would not compile otherwise;
    this catches a field added to the header and forgotten in the transliteration)."""
    body = re.search(r"pub fn desc\(&self\) -> RaycaSceneDesc \{\s*RaycaSceneDesc \{(.*?)\n        \}", SHIM, flags=re.S).group(1)
    set_fields = re.findall(r"^\s*(\w+):", body, flags=re.M)
    assert set_fields == [f[0] for f in c_structs()["RaycaSceneDesc"]]

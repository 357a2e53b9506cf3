"""The Rust bindings under shim/ (and the Rust snippets of INTEGRATION.md) declare the SAME structs and prototypes as include/zkmi355.h.

Nothing in the build image compiles Rust, so a `#[repr(C)]` struct that falls behind the header would only show up as an out-of-bounds read on the first machine
with cargo (round 3: ZkPlonkPkHost had not followed zk_plonk_pk_host's four shard fields).  This test parses both sides and compares

  * every `#[repr(C)] struct` with the C struct of the same name (ZkPlonkPkHost <-> zk_plonk_pk_host): field names, order and types;
  * every function of every `extern "C" { .. }` block with the header's prototype: existence, arity, argument and return types;
  * every `type X = extern "C" fn(..)` with the header's function-pointer typedef;
  * the binding's ZK_ABI_VERSION constant with the header's.

Types are compared after mapping both sides to one spelling: (pointer chain with constness, base type)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zkmi355.h")
RCCL_HEADER = os.path.join(ROOT, "include", "zkmi355_rccl.h")      # the optional RCCL adapter (libzkmi355_rccl.so): shim/halo2_proofs_mi355x/src/rccl.rs binds it


def header_text():
    return open(HEADER).read() + "\n" + open(RCCL_HEADER).read()

C_BASE = {"uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "int": "c_int", "size_t": "usize", "double": "f64", "void": "c_void", "char": "c_char"}


def snake(name):
    return re.sub(r"(?<!^)([A-Z])", r"_\1", name).lower()


# ---- the C side ----------------------------------------------------------------------------------------------------------------------------------------
def c_type(decl_type, array=False):
    """'const void* const*' -> ('c_void', ['const', 'const']) : constness of each pointer level's TARGET, innermost first"""
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", decl_type)
    base, ptrs, pending_const = None, [], False
    for t in toks:
        if t == "const":
            pending_const = True
        elif t == "*":
            ptrs.append("const" if pending_const else "mut")
            pending_const = False
        elif t in ("struct", "unsigned"):
            continue
        else:
            assert base is None, decl_type
            base = C_BASE.get(t, t)
    if array:                                           # `uint32_t counts[9]` as a parameter is a pointer to mutable elements
        ptrs.append("const" if pending_const else "mut")
    return base, ptrs


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [p.strip() for p in out]


def c_param(p):
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[0-9]*\])?$", p.strip())
    assert m, p
    return m.group(2), c_type(m.group(1), array=bool(m.group(3)))


def parse_header(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    structs, fnptrs, protos = {}, {}, {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first, *rest = [d.strip() for d in decl.split(",")]
            name, ty = c_param(first)
            fields.append((name, ty))
            for r in rest:                               # `uint32_t k, extended_k;` — further declarators share the base type (and carry their own stars)
                stars = r.count("*")
                fields.append((r.replace("*", "").strip(), (ty[0], ["mut"] * stars)))
        structs[m.group(1)] = fields
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", text, flags=re.S):
        fnptrs[m.group(2)] = (c_type(m.group(1)), [c_param(p)[1] for p in split_params(" ".join(m.group(3).split()))])
    text = re.sub(r"typedef[^;]*;", "", text)
    for m in re.finditer(r"([\w\s\*]+?)\b(zk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = " ".join(m.group(3).split())
        args = [] if params in ("", "void") else [c_param(p)[1] for p in split_params(params)]
        protos[m.group(2)] = (c_type(m.group(1)), args)
    return structs, fnptrs, protos


# ---- the Rust side -------------------------------------------------------------------------------------------------------------------------------------
RUST_BASE = {"c_void": "c_void", "c_int": "c_int", "c_char": "c_char", "u32": "u32", "u64": "u64", "u8": "u8", "usize": "usize", "f64": "f64"}


def rust_type(t):
    t = t.strip()
    m = re.match(r"^Option<(.*)>$", t)
    if m:                                               # Option<extern "C" fn> / Option<ZkFn> has the layout of the nullable C function pointer
        t = m.group(1).strip()
    ptrs = []
    while True:
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        if not m:
            break
        ptrs.append(m.group(1))
        t = m.group(2).strip()
    base = RUST_BASE.get(t, snake(t) if re.match(r"^Zk[A-Z]", t) else t)
    return base, list(reversed(ptrs))                   # innermost first, like c_type


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def parse_rust(text):
    text = strip_rust_comments(text)
    structs, externs, fntypes, consts = {}, {}, {}, {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub\s+struct\s+(\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for f in split_params(m.group(2)):
            f = " ".join(f.split())
            if not f:
                continue
            fm = re.match(r"^(?:pub(?:\([a-z]+\))?\s+)?(\w+)\s*:\s*(.*)$", f)
            assert fm, f
            fields.append((fm.group(1), rust_type(fm.group(2))))
        structs[m.group(1)] = fields
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\s*\}', text, flags=re.S):
        for m in re.finditer(r"(?:pub\s+)?fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk.group(1), flags=re.S):
            args = [rust_type(p.split(":", 1)[1]) for p in split_params(" ".join(m.group(2).split())) if p]
            externs[m.group(1)] = (rust_type(m.group(3)) if m.group(3) else ("c_void", []), args)
    for m in re.finditer(r'type\s+(\w+)\s*=\s*extern\s+"C"\s+fn\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;', text, flags=re.S):
        args = [rust_type(p.split(":", 1)[1] if ":" in p else p) for p in split_params(" ".join(m.group(2).split())) if p]
        fntypes[m.group(1)] = (rust_type(m.group(3)) if m.group(3) else ("c_void", []), args)
    for m in re.finditer(r"const\s+(ZK_\w+)\s*:\s*u32\s*=\s*(\d+)\s*;", text):
        consts[m.group(1)] = int(m.group(2))
    return structs, externs, fntypes, consts


def rust_sources():
    out = []
    for dp, _, files in os.walk(os.path.join(ROOT, "shim")):
        for f in sorted(files):
            if f.endswith(".rs"):
                out.append(os.path.join(dp, f))
    return sorted(out)


def markdown_rust_blocks(path):
    return "\n".join(m.group(1) for m in re.finditer(r"```rust\n(.*?)```", open(path).read(), flags=re.S))


def diff_against_header(rust_text, where, header_text=None):
    """every mismatch between the declarations in rust_text and the header, as strings (empty = in step)"""
    structs, fnptrs, protos = parse_header(header_text if header_text is not None else globals()["header_text"]())
    r_structs, r_externs, r_fntypes, r_consts = parse_rust(rust_text)
    bad = []
    for name, fields in r_structs.items():
        cname = snake(name)
        if cname in ("zk_ctx", "zk_rccl_comm"):          # opaque on both sides
            continue
        if cname not in structs:
            bad.append(f"{where}: #[repr(C)] struct {name} has no C struct {cname} in the header")
            continue
        want = structs[cname]
        if [n for n, _ in fields] != [n for n, _ in want]:
            bad.append(f"{where}: {name} fields {[n for n, _ in fields]} != {cname} fields {[n for n, _ in want]}")
            continue
        for (n, got), (_, exp) in zip(fields, want):
            exp_base = exp[0] if exp[0] not in fnptrs else exp[0]
            if (got[0], got[1]) != (exp_base, exp[1]):
                bad.append(f"{where}: {name}.{n}: Rust {got} vs C {exp}")
    for fn, (ret, args) in r_externs.items():
        if fn not in protos:
            bad.append(f"{where}: extern fn {fn} is not declared in the header")
            continue
        cret, cargs = protos[fn]
        if len(args) != len(cargs):
            bad.append(f"{where}: {fn}: {len(args)} arguments, the header has {len(cargs)}")
            continue
        if ret != cret and not (cret == ("c_void", []) and ret == ("c_void", [])):
            bad.append(f"{where}: {fn}: returns {ret}, the header says {cret}")
        for i, (got, exp) in enumerate(zip(args, cargs)):
            if got != exp:
                bad.append(f"{where}: {fn} argument {i}: Rust {got} vs C {exp}")
    for name, (ret, args) in r_fntypes.items():
        cname = snake(name)
        if cname not in fnptrs:
            bad.append(f"{where}: fn type {name} has no typedef {cname} in the header")
            continue
        if (ret, args) != fnptrs[cname]:
            bad.append(f"{where}: fn type {name}: Rust {(ret, args)} vs C {fnptrs[cname]}")
    m = re.search(r"#define\s+ZK_ABI_VERSION\s+(\d+)", header_text if header_text is not None else open(HEADER).read())
    if "ZK_ABI_VERSION" in r_consts and int(m.group(1)) != r_consts["ZK_ABI_VERSION"]:
        bad.append(f"{where}: ZK_ABI_VERSION {r_consts['ZK_ABI_VERSION']} vs the header's {m.group(1)}")
    return bad


# ---- tests ---------------------------------------------------------------------------------------------------------------------------------------------
def test_header_parses_completely():
    structs, fnptrs, protos = parse_header(open(HEADER).read())
    assert set(structs) == {"zk_quotient_args", "zk_plonk_pk_desc", "zk_plonk_pk_host"}
    assert set(fnptrs) == {"zk_allgather_fn", "zk_rng_fn"}
    from test_capi_symbols import declared_symbols
    assert sorted(protos) == [s for s in declared_symbols() if s not in fnptrs], "a prototype of the header escaped the parser"
    for name, fields in structs.items():
        assert fields[0] == ("struct_size", ("u32", [])), f"{name} must start with uint32_t struct_size (ABI versioning)"


def test_rust_bindings_match_the_header():
    seen_structs, seen_fns, bad = set(), set(), []
    for path in rust_sources():
        text = open(path).read()
        bad += diff_against_header(text, os.path.relpath(path, ROOT))
        s, e, _, _ = parse_rust(text)
        seen_structs |= set(s)
        seen_fns |= set(e)
    bad += diff_against_header(markdown_rust_blocks(os.path.join(ROOT, "INTEGRATION.md")), "INTEGRATION.md")
    assert not bad, "\n".join(bad)
    # the parser did see the binding (an empty comparison proves nothing)
    assert {"ZkQuotientArgs", "ZkPlonkPkHost"} <= seen_structs
    assert {"zk_plonk_pk_build", "zk_plonk_prove", "zk_msm_batch", "zk_ntt", "zk_abi_version", "zk_abi_struct_size"} <= seen_fns
    # the RCCL adapter: every function of its header is bound, with the header's types
    _, _, rccl_protos = parse_header(open(RCCL_HEADER).read())
    assert set(rccl_protos) == {f for f in seen_fns if f.startswith("zk_rccl_")} and len(rccl_protos) == 9, (sorted(rccl_protos), sorted(f for f in seen_fns if f.startswith("zk_rccl_")))


def test_the_round3_drift_is_caught():
    """the defect this test exists for: the Rust struct without the header's trailing shard fields (and without struct_size) must be reported"""
    path = os.path.join(ROOT, "shim", "halo2_proofs_mi355x", "src", "pk_desc.rs")
    text = open(path).read()
    cut = re.sub(r"\n\s*// one proof over several GPUs.*?allgather_user: \*mut c_void,", "", text, count=1, flags=re.S)
    assert cut != text
    bad = diff_against_header(cut, "pk_desc.rs (shard fields removed)")
    assert any("ZkPlonkPkHost fields" in b for b in bad), bad
    # a type slip in place (u32 -> u64) and a missing argument are reported too
    assert any("k:" in b or ".k" in b for b in diff_against_header(text.replace("pub k: u32,", "pub k: u64,", 1), "pk_desc.rs (k widened)"))
    fewer = text.replace("srs_g: u64, srs_g_lagrange: u64, pk: *mut u64", "srs_g: u64, pk: *mut u64", 1)
    assert any("zk_plonk_pk_build" in b and "arguments" in b for b in diff_against_header(fewer, "pk_desc.rs (argument dropped)"))


def test_patches_only_call_functions_the_binding_defines():
    """`crate::mi355x::name(` / `crate::pk_desc::name(` / `crate::create_proof_native::name` in a .patch or .rs must be a `pub fn` / `pub(crate) fn` of that module"""
    defined = {}
    for mod, rel in (("mi355x", "shim/halo2_proofs_mi355x/src/mi355x.rs"), ("pk_desc", "shim/halo2_proofs_mi355x/src/pk_desc.rs"),
                     ("create_proof_native", "shim/halo2_proofs_mi355x/src/create_proof_native.rs")):
        defined[mod] = set(re.findall(r"pub(?:\(crate\))?\s+(?:unsafe\s+)?fn\s+(\w+)", strip_rust_comments(open(os.path.join(ROOT, rel)).read())))
    defined["mi355x"] |= set(re.findall(r"pub(?:\(crate\))?\s+fn\s+(\w+)", strip_rust_comments(open(os.path.join(ROOT, "shim/halo2_axiom_mi355x/src/mi355x.rs")).read())))
    missing = []
    for dp, _, files in os.walk(os.path.join(ROOT, "shim")):
        for f in files:
            if f.endswith((".patch", ".rs")):
                for mod, fn in re.findall(r"crate::(mi355x|pk_desc|create_proof_native)::([a-z_0-9]+)\s*(?:::<[^(]*>)?\(", open(os.path.join(dp, f)).read()):
                    if fn not in defined[mod]:
                        missing.append(f"{f}: crate::{mod}::{fn} is not defined")
    assert not missing, missing


def test_library_reports_the_header_sizes(built):
    """zk_abi_version / zk_abi_struct_size of the built library against the header and the ctypes mirrors (x86-64 SysV layout computed here from the parsed header)"""
    import ctypes as C
    import zk_dcap_verifier_amd as z
    from zk_dcap_verifier_amd import _lib
    from zk_dcap_verifier_amd.plonk import native
    lib = C.CDLL(z.LIB_PATH)
    lib.zk_abi_version.restype = C.c_uint32
    lib.zk_abi_struct_size.restype = C.c_uint32
    hdr = open(HEADER).read()
    assert lib.zk_abi_version() == int(re.search(r"#define\s+ZK_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION
    structs, fnptrs, _ = parse_header(hdr)

    def size_of(fields):
        off, align = 0, 1
        for _, (base, ptrs) in fields:
            sz = 8 if ptrs or base in fnptrs or base in ("u64", "usize", "f64") else 4
            off = (off + sz - 1) // sz * sz + sz
            align = max(align, sz)
        return (off + align - 1) // align * align
    for name, fields in structs.items():
        assert lib.zk_abi_struct_size(name.encode()) == size_of(fields), name
    assert lib.zk_abi_struct_size(b"zk_quotient_args") == C.sizeof(_lib.QuotientArgs)
    assert lib.zk_abi_struct_size(b"zk_plonk_pk_desc") == C.sizeof(native.PkDesc)
    assert [n for n, _ in structs["zk_quotient_args"]] == [n for n, _ in _lib.QuotientArgs._fields_]
    assert [n for n, _ in structs["zk_plonk_pk_desc"]] == [n for n, _ in native.PkDesc._fields_]
    assert lib.zk_abi_struct_size(b"nonsense") == 0

"""rust/src/modules/hip_ffi.rs declares every function of include/stark_hip.h with the same arguments,
type by type, and the same return type (the Rust side cannot be compiled where this runs; this keeps the
two from drifting: a usize for a uint64_t, a *const for a *mut or an i32 for a u32 would otherwise first show
as memory corruption on the first machine with cargo), and the new bodies only call functions that are declared."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip(text, pattern):
    return re.sub(pattern, "", text, flags=re.S)


def _header():
    text = _strip(open(os.path.join(ROOT, "include", "stark_hip.h")).read(), r"/\*.*?\*/")
    text = re.sub(r"#.*", "", text)
    out = {}
    for name, args in re.findall(r"\b(stark_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        if name == "stark_transcript_cb":
            continue
        args = " ".join(args.split())
        out[name] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust():
    text = _strip(open(os.path.join(ROOT, "rust", "src", "modules", "hip_ffi.rs")).read(), r"//[^\n]*")
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for name, args in re.findall(r"pub fn (stark_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", block, flags=re.S):
        args = " ".join(args.split())
        out[name] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_every_entry_point_is_declared_with_the_same_arity():
    h, r = _header(), _rust()
    assert sorted(h) == sorted(r)
    assert len(h) >= 50
    for name in h:
        assert h[name] == r[name], f"{name}: header has {h[name]} arguments, hip_ffi.rs {r[name]}"


C_BASE = {"int32_t": "i32", "uint32_t": "u32", "size_t": "usize", "uint64_t": "u64", "uint8_t": "u8", "void": "c_void",
          "char": "c_char", "stark_ctx": "StarkCtx", "stark_tree": "StarkTree", "stark_fri_layers": "StarkFriLayers"}


def c_type_to_rust(decl: str, is_return: bool = False) -> str:
    """One C parameter (or return type) as the Rust FFI type that has the same ABI."""
    decl = " ".join(decl.replace("*", " * ").replace("[", " [").split())
    array = bool(re.search(r"\[\d*\]", decl))
    decl = re.sub(r"\[\d*\]", "", decl).strip()
    toks = decl.split()
    if toks == ["stark_transcript_cb", toks[-1]] and not is_return:
        return "TranscriptCb"
    # drop the parameter name: the last identifier, unless the declaration is a bare type
    if not is_return and re.fullmatch(r"[A-Za-z_]\w*", toks[-1]) and toks[-1] not in C_BASE:
        toks = toks[:-1]
    # read left to right: [const] base, then any number of "* [const]"
    i, const = 0, False
    if toks[i] == "const":
        const, i = True, i + 1
    base = C_BASE[toks[i]]
    i += 1
    if i < len(toks) and toks[i] == "const":
        const, i = True, i + 1
    levels = []                                            # constness of what each pointer level points TO
    pointee_const = const
    while i < len(toks):
        assert toks[i] == "*", decl
        levels.append(pointee_const)
        i += 1
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const, i = True, i + 1
    if array:
        levels.append(pointee_const if levels else const)
    out = base
    for c in levels:
        out = ("*const " if c else "*mut ") + out
    return out


def _header_types():
    text = _strip(open(os.path.join(ROOT, "include", "stark_hip.h")).read(), r"/\*.*?\*/")
    text = re.sub(r"#.*", "", text)
    out = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w \*]*?)\b(stark_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        if name == "stark_transcript_cb" or "typedef" in ret:
            continue
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [c_type_to_rust(a) for a in args.split(",")]
        ret = " ".join(ret.split())
        out[name] = (params, None if ret == "void" else c_type_to_rust(ret, is_return=True))
    return out


def _rust_types():
    text = _strip(open(os.path.join(ROOT, "rust", "src", "modules", "hip_ffi.rs")).read(), r"//[^\n]*")
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for name, args, ret in re.findall(r"pub fn (stark_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*([^;]*?))?\s*;", block, flags=re.S):
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        out[name] = (params, " ".join(ret.split()) or None)
    return out


def test_c_type_translation():
    assert c_type_to_rust("stark_ctx* ctx") == "*mut StarkCtx"
    assert c_type_to_rust("const stark_tree* tree") == "*const StarkTree"
    assert c_type_to_rust("stark_tree** tree_out") == "*mut *mut StarkTree"
    assert c_type_to_rust("const uint64_t offset[2]") == "*const u64"
    assert c_type_to_rust("uint8_t root[32]") == "*mut u8"
    assert c_type_to_rust("const uint64_t* const* srcs_dev") == "*const *const u64"
    assert c_type_to_rust("void** dptr") == "*mut *mut c_void"
    assert c_type_to_rust("const void* src") == "*const c_void"
    assert c_type_to_rust("size_t n") == "usize" and c_type_to_rust("int32_t inverse") == "i32"
    assert c_type_to_rust("stark_transcript_cb cb") == "TranscriptCb"
    assert c_type_to_rust("const char*", is_return=True) == "*const c_char"


def test_every_entry_point_has_the_same_argument_and_return_types():
    h, r = _header_types(), _rust_types()
    assert sorted(h) == sorted(r)
    for name in h:
        assert h[name][0] == r[name][0], f"{name}: header arguments {h[name][0]}, hip_ffi.rs {r[name][0]}"
        assert h[name][1] == r[name][1], f"{name}: header returns {h[name][1]}, hip_ffi.rs {r[name][1]}"


def test_transcript_callback_type():
    text = open(os.path.join(ROOT, "rust", "src", "modules", "hip_ffi.rs")).read()
    m = re.search(r"pub type TranscriptCb\s*=\s*extern \"C\" fn\(([^)]*)\)\s*->\s*(\w+);", text, flags=re.S)
    assert m, "TranscriptCb not found"
    rust = [" ".join(a.split(":", 1)[1].split()) for a in m.group(1).split(",") if a.strip()]
    header = _strip(open(os.path.join(ROOT, "include", "stark_hip.h")).read(), r"/\*.*?\*/")
    c = re.search(r"typedef\s+int32_t\s*\(\*stark_transcript_cb\)\s*\(([^)]*)\)", header, flags=re.S)
    assert c, "stark_transcript_cb not found"
    assert [c_type_to_rust(a) for a in " ".join(c.group(1).split()).split(",")] == rust and m.group(2) == "i32"


def test_new_bodies_call_declared_functions_only():
    r = _rust()
    for f in ("hip_backend.rs",):
        text = open(os.path.join(ROOT, "rust", "src", "modules", f)).read()
        used = set(re.findall(r"\b(stark_[a-z0-9_]+)\s*\(", text))
        assert used and used <= set(r), used - set(r)
    ab = open(os.path.join(ROOT, "rust", "build.rs")).read()
    built = set(re.findall(r'"([a-z_]+\.hip)"', ab))
    assert built == {f for f in os.listdir(os.path.join(ROOT, "stark_amd", "csrc")) if f.endswith(".hip")}

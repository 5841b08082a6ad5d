"""rust/src/modules/hip_ffi.rs declares every function of include/stark_hip.h with the same number
of arguments (the Rust side cannot be compiled where this runs; this keeps the two from drifting),
and the new bodies only call functions that are declared."""
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


def test_new_bodies_call_declared_functions_only():
    r = _rust()
    for f in ("hip_backend.rs",):
        text = open(os.path.join(ROOT, "rust", "src", "modules", f)).read()
        used = set(re.findall(r"\b(stark_[a-z0-9_]+)\s*\(", text))
        assert used and used <= set(r), used - set(r)
    ab = open(os.path.join(ROOT, "rust", "build.rs")).read()
    built = set(re.findall(r'"([a-z_]+\.hip)"', ab))
    assert built == {f for f in os.listdir(os.path.join(ROOT, "stark_amd", "csrc")) if f.endswith(".hip")}

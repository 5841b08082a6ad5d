"""CPU-only checks of the C-ABI library: it loads, exports every symbol that
include/stark_hip.h declares, and the ctypes table mirrors the header."""
import os
import re

from stark_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "stark_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(stark_[a-z0-9_]+)\s*\(", text)
    # drop the callback typedef name
    return sorted(set(n for n in names if n != "stark_transcript_cb"))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in stark_hip.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _header_functions()


def test_num_rounds_is_host_only():
    lib = _lib.load()
    assert lib.stark_fri_num_rounds(1 << 24, 4, 17) == 18
    assert lib.stark_fri_num_rounds(256, 4, 17) == 2
    assert lib.stark_fri_num_rounds(4, 4, 1) == 0


def test_header_is_plain_c_and_links(tmp_path):
    """The header compiles as C99 (no C++ in the boundary) and a C program that takes the address
    of every declared entry point links against the library."""
    import subprocess
    names = _header_functions()
    src = tmp_path / "abi.c"
    src.write_text('#include "stark_hip.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n    fn f[] = {\n'
                   + "".join(f"        (fn)&{n},\n" for n in names)
                   + '    };\n    printf("%d\\n", (int)(sizeof f / sizeof f[0]));\n'
                   '    printf("%d\\n", (int)stark_fri_num_rounds(256, 4, 17));\n    return 0;\n}\n')
    exe = tmp_path / "abi"
    libdir = os.path.join(ROOT, "stark_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lstark_hip", f"-Wl,-rpath,{libdir}"])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert out == [str(len(names)), "2"]

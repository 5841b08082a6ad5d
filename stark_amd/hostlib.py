"""ctypes access to the C++ host mirror of the reference API
(stark_amd/host/libstark_host.so: FieldElement, ProofStream, NTT, Polynomial,
Merkle, Fri above the C ABI)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libstark_host.so")
if not os.path.exists(PATH):
    raise ImportError(f"{PATH} is missing: build it with `make -C stark_amd/host` (needs libstark_hip.so first)")

from ._lib import preload_hip_runtime  # noqa: E402

preload_hip_runtime()
lib = C.CDLL(PATH)
lib.sh_last_panic.restype = C.c_char_p
lib.sh_ps_new.restype = C.c_void_p
lib.sh_ps_deserialize.restype = C.c_void_p
lib.sh_context.restype = C.c_void_p
for _name in ("sh_fe_to_string", "sh_bincode_fe", "sh_ps_count", "sh_ps_object", "sh_ps_serialize"):
    getattr(lib, _name).restype = C.c_size_t
_vp = C.c_void_p


class HostPanic(RuntimeError):
    """The mirror raised what the reference would panic with."""


def chk(rc):
    if rc < 0:
        raise HostPanic(lib.sh_last_panic().decode())
    return rc


def fe(v):
    return (C.c_uint64 * 2)(v & (2**64 - 1), v >> 64)


def val(a):
    return int(a[0]) | (int(a[1]) << 64)


def arr(vals):
    buf = b"".join(int(v).to_bytes(16, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).copy()


def ints(a):
    b = np.ascontiguousarray(a).tobytes()
    return [int.from_bytes(b[i:i + 16], "little") for i in range(0, len(b), 16)]


def p(a):
    """Pointer to a numpy array's data; the returned object keeps the array alive."""
    return a.ctypes.data_as(_vp)


def fe_op(op, a, b=None):
    out = (C.c_uint64 * 2)()
    chk(lib.sh_fe_op(op, fe(a), fe(b) if b is not None else None, out))
    return val(out)


def primitive_nth_root(n):
    out = (C.c_uint64 * 2)()
    chk(lib.sh_primitive_nth_root(fe(n), out))
    return val(out)


def flatten(leaves):
    flat = np.frombuffer(b"".join(leaves) or b"\0", dtype=np.uint8).copy()
    offs = np.zeros(len(leaves) + 1, dtype=np.uint64)
    if leaves:
        offs[1:] = np.cumsum([len(l) for l in leaves])
    return flat, offs


class Stream:
    """ProofStream (proof_stream.rs:8-69) held by the C++ mirror."""

    def __init__(self, handle=None):
        self.h = None
        self.h = _vp(handle) if handle else _vp(lib.sh_ps_new())

    def __del__(self, _lib=lib):                 # the module's globals may be gone when the interpreter shuts down
        if getattr(self, "h", None) and _lib is not None:
            _lib.sh_ps_free(self.h)
            self.h = None

    def push(self, s):
        b = s.encode()
        lib.sh_ps_push(self.h, b, C.c_size_t(len(b)))

    @property
    def objects(self):
        out = []
        for i in range(lib.sh_ps_count(self.h)):
            n = lib.sh_ps_object(self.h, C.c_size_t(i), None)
            buf = C.create_string_buffer(n)
            lib.sh_ps_object(self.h, C.c_size_t(i), buf)
            out.append(buf.raw[:n].decode())
        return out

    def serialize(self):
        n = lib.sh_ps_serialize(self.h, None)
        buf = (C.c_uint8 * n)()
        lib.sh_ps_serialize(self.h, buf)
        return bytes(buf)

    def fiat_shamir(self, nbytes=32, verifier=False):
        out = (C.c_uint8 * nbytes)()
        lib.sh_ps_fiat_shamir(self.h, int(verifier), out, C.c_size_t(nbytes))
        return bytes(out)

    @staticmethod
    def deserialize(data):
        b = (C.c_uint8 * len(data)).from_buffer_copy(data)
        h = lib.sh_ps_deserialize(b, C.c_size_t(len(data)))
        if not h:
            raise HostPanic(lib.sh_last_panic().decode())
        return Stream(h)

    @staticmethod
    def from_objects(objs):
        s = Stream()
        for o in objs:
            s.push(o)
        return s


def fri_commit_resident(dev_ptr, n, offset, omega, expansion, ncol, stream):
    """Fri::commit (fri.rs:100-152) of a packed codeword already on the device; the
    transcript callback is the mirror's native ProofStream.  Returns the layer count."""
    return chk(lib.sh_fri_commit_resident(_vp(dev_ptr), C.c_size_t(n), fe(offset), fe(omega), C.c_size_t(expansion),
                                          C.c_size_t(ncol), stream.h))


def fri_prove_resident(dev_ptr, n, offset, omega, expansion, ncol, stream):
    """Fri::prove (fri.rs:212-254) of a codeword resident on the device; returns the top-level indices."""
    out = (C.c_uint64 * (2 * ncol))()
    k = chk(lib.sh_fri_prove_resident(_vp(dev_ptr), C.c_size_t(n), fe(offset), fe(omega), C.c_size_t(expansion),
                                      C.c_size_t(ncol), stream.h, out))
    return [int(out[i]) for i in range(k)]


def fri_verify(offset, omega, n, expansion, ncol, stream, max_points=4096):
    idx = np.zeros(max_points, dtype=np.uint64)
    vals = np.zeros((max_points, 2), dtype=np.uint64)
    npts = C.c_size_t()
    rc = chk(lib.sh_fri_verify(fe(offset), fe(omega), C.c_size_t(n), C.c_size_t(expansion), C.c_size_t(ncol), stream.h,
                               p(idx), p(vals), C.byref(npts)))
    k = npts.value
    return bool(rc), list(zip([int(i) for i in idx[:k]], ints(vals[:k])))

"""ctypes bindings for the C restatement (oracle/stark_oracle.c).

Test infrastructure: used by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "_build", "libstark_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    u64p = C.POINTER(C.c_uint64)
    u8p = C.POINTER(C.c_uint8)
    sz = C.c_size_t
    sigs = {
        "so_fe_mul_vec": (None, [u64p, u64p, u64p, sz]),
        "so_fe_add": (None, [u64p, u64p, u64p]),
        "so_fe_sub": (None, [u64p, u64p, u64p]),
        "so_fe_mul": (None, [u64p, u64p, u64p]),
        "so_fe_pow": (None, [u64p, u64p, u64p]),
        "so_fe_inverse": (None, [u64p, u64p]),
        "so_primitive_nth_root": (C.c_int, [C.c_uint64, u64p]),
        "so_fe_sample": (None, [u8p, sz, u64p]),
        "so_ntt": (C.c_int, [u64p, sz, C.c_int]),
        "so_ntt_evaluate": (C.c_int, [u64p, sz, sz, u64p]),
        "so_coset_lde": (C.c_int, [u64p, sz, sz, u64p, u64p]),
        "so_coset_lde_horner": (C.c_int, [u64p, sz, sz, u64p, u64p]),
        "so_blake2b": (None, [u8p, sz, u8p, sz]),
        "so_shake256": (None, [u8p, sz, u8p, sz]),
        "so_bincode_fe": (sz, [u64p, u8p]),
        "so_merkle_tree_fe": (C.c_int, [u64p, sz, u8p]),
        "so_merkle_commit_fe": (C.c_int, [u64p, sz, u8p]),
        "so_merkle_tree_bytes": (C.c_int, [u8p, u64p, sz, u8p]),
        "so_merkle_commit_bytes": (C.c_int, [u8p, u64p, sz, u8p]),
        "so_merkle_open": (C.c_int, [u8p, sz, sz, u8p]),
        "so_merkle_verify": (C.c_int, [u8p, sz, u8p, sz, u8p, sz]),
        "so_ps_new": (C.c_void_p, []),
        "so_ps_free": (None, [C.c_void_p]),
        "so_ps_push": (None, [C.c_void_p, C.c_char_p, sz]),
        "so_ps_count": (sz, [C.c_void_p]),
        "so_ps_object_len": (sz, [C.c_void_p, sz]),
        "so_ps_object": (C.c_void_p, [C.c_void_p, sz]),
        "so_ps_serialize": (sz, [C.c_void_p, sz, u8p]),
        "so_ps_fiat_shamir": (None, [C.c_void_p, u8p, sz]),
        "so_fri_num_rounds": (sz, [sz, sz, sz]),
        "so_fri_fold": (None, [u64p, sz, u64p, u64p, u64p, u64p]),
        "so_fri_commit": (C.c_int, [u64p, sz, u64p, u64p, sz, sz, C.c_void_p, u64p, u8p, u64p]),
        "so_fri_sample_indices": (C.c_int, [u8p, sz, C.c_uint64, C.c_uint64, sz, u64p]),
        "so_splitmix_field_vector": (None, [u64p, sz, C.c_uint64]),
        "so_set_threads": (None, [C.c_int]),
        "so_get_threads": (C.c_int, []),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def p64(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def p8(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


def fe(v: int) -> np.ndarray:
    return np.array([v & (2**64 - 1), v >> 64], dtype=np.uint64)


def to_int(a: np.ndarray) -> int:
    return int(a[0]) | (int(a[1]) << 64)


def from_ints(vals) -> np.ndarray:
    """list of ints -> (n,2) uint64 array {lo,hi}."""
    buf = b"".join(int(v).to_bytes(16, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).copy()


def to_ints(a: np.ndarray):
    b = np.ascontiguousarray(a).tobytes()
    return [int.from_bytes(b[i:i + 16], "little") for i in range(0, len(b), 16)]


class threads:
    """with threads(k): the bulk loops of the restatement (hashing, fold, butterflies) run on k threads.
    The values do not depend on k; only the all-cores CPU baseline and its test use more than one."""

    def __init__(self, k: int):
        self.k = k

    def __enter__(self):
        self.prev = lib.so_get_threads()
        lib.so_set_threads(self.k)
        return self

    def __exit__(self, *exc):
        lib.so_set_threads(self.prev)


def splitmix(n: int, seed: int = 0x5741524B) -> np.ndarray:
    out = np.empty((n, 2), dtype=np.uint64)
    lib.so_splitmix_field_vector(p64(out), n, seed)
    return out


def ntt(data: np.ndarray, inverse: bool = False) -> np.ndarray:
    out = np.ascontiguousarray(data).copy()
    rc = lib.so_ntt(p64(out), out.shape[0], int(inverse))
    if rc:
        raise ValueError(f"so_ntt rc={rc}")
    return out


def coset_lde(coeffs: np.ndarray, n_out: int, offset: int) -> np.ndarray:
    out = np.empty((n_out, 2), dtype=np.uint64)
    rc = lib.so_coset_lde(p64(np.ascontiguousarray(coeffs)), coeffs.shape[0], n_out, p64(fe(offset)), p64(out))
    if rc:
        raise ValueError(f"so_coset_lde rc={rc}")
    return out


def blake2b(data: bytes, outlen: int = 32) -> bytes:
    inp = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, dtype=np.uint8)
    out = np.empty(outlen, dtype=np.uint8)
    lib.so_blake2b(p8(inp), len(data), p8(out), outlen)
    return out.tobytes()


def shake256(data: bytes, outlen: int) -> bytes:
    inp = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, dtype=np.uint8)
    out = np.empty(outlen, dtype=np.uint8)
    lib.so_shake256(p8(inp), len(data), p8(out), outlen)
    return out.tobytes()


def bincode_fe(v: int) -> bytes:
    out = np.empty(25, dtype=np.uint8)
    n = lib.so_bincode_fe(p64(fe(v)), p8(out))
    return out[:n].tobytes()


def merkle_tree_fe(elems: np.ndarray) -> np.ndarray:
    n = elems.shape[0]
    tree = np.empty((2 * n - 1, 32), dtype=np.uint8)
    rc = lib.so_merkle_tree_fe(p64(np.ascontiguousarray(elems)), n, p8(tree))
    if rc:
        raise ValueError(f"so_merkle_tree_fe rc={rc}")
    return tree


def merkle_commit_fe(elems: np.ndarray) -> bytes:
    root = np.empty(32, dtype=np.uint8)
    rc = lib.so_merkle_commit_fe(p64(np.ascontiguousarray(elems)), elems.shape[0], p8(root))
    if rc:
        raise ValueError(f"so_merkle_commit_fe rc={rc}")
    return root.tobytes()


def _flatten(leaves):
    flat = np.frombuffer(b"".join(leaves) or b"\0", dtype=np.uint8).copy()
    offs = np.zeros(len(leaves) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(l) for l in leaves])
    return flat, offs


def merkle_commit_bytes(leaves) -> bytes:
    flat, offs = _flatten(leaves)
    root = np.empty(32, dtype=np.uint8)
    rc = lib.so_merkle_commit_bytes(p8(flat), p64(offs), len(leaves), p8(root))
    if rc:
        raise ValueError(f"so_merkle_commit_bytes rc={rc}")
    return root.tobytes()


def merkle_open(tree: np.ndarray, index: int):
    n = (tree.shape[0] + 1) // 2
    path = np.empty((64, 32), dtype=np.uint8)
    k = lib.so_merkle_open(p8(tree), n, index, p8(path))
    if k < 0:
        raise ValueError(f"so_merkle_open rc={k}")
    return [path[i].tobytes() for i in range(k)]


def fri_fold(cw: np.ndarray, alpha: int, offset: int, omega: int) -> np.ndarray:
    n = cw.shape[0]
    out = np.empty((n // 2, 2), dtype=np.uint64)
    lib.so_fri_fold(p64(np.ascontiguousarray(cw)), n, p64(fe(alpha)), p64(fe(offset)), p64(fe(omega)), p64(out))
    return out


class Stream:
    """C-side ProofStream (proof_stream.rs)."""

    def __init__(self):
        self.h = lib.so_ps_new()

    def __del__(self):
        lib.so_ps_free(self.h)

    def push(self, s: str):
        b = s.encode()
        lib.so_ps_push(self.h, b, len(b))

    @property
    def objects(self):
        out = []
        for i in range(lib.so_ps_count(self.h)):
            n = lib.so_ps_object_len(self.h, i)
            out.append(C.string_at(lib.so_ps_object(self.h, i), n).decode())
        return out

    def serialize(self) -> bytes:
        cnt = lib.so_ps_count(self.h)
        n = lib.so_ps_serialize(self.h, cnt, None)
        buf = np.empty(n, dtype=np.uint8)
        lib.so_ps_serialize(self.h, cnt, p8(buf))
        return buf.tobytes()

    def fiat_shamir(self, nbytes: int = 32) -> bytes:
        out = np.empty(nbytes, dtype=np.uint8)
        lib.so_ps_fiat_shamir(self.h, p8(out), nbytes)
        return out.tobytes()


def fri_commit(codeword: np.ndarray, offset: int, omega: int, expansion: int, ncol: int, stream: Stream = None):
    """Returns (layers list of arrays, roots list of bytes, alphas list of int, stream)."""
    n = codeword.shape[0]
    rounds = lib.so_fri_num_rounds(n, expansion, ncol)
    if rounds == 0:
        raise ValueError("Cannot do FRI with less than 1 round")
    stream = stream or Stream()
    total = sum(n >> r for r in range(rounds))
    layers = np.empty((total, 2), dtype=np.uint64)
    roots = np.empty((rounds, 32), dtype=np.uint8)
    alphas = np.zeros((max(rounds - 1, 1), 2), dtype=np.uint64)
    rc = lib.so_fri_commit(p64(np.ascontiguousarray(codeword)), n, p64(fe(offset)), p64(fe(omega)),
                           expansion, ncol, stream.h, p64(layers), p8(roots), p64(alphas))
    if rc < 0:
        raise ValueError(f"so_fri_commit rc={rc}")
    outl, off = [], 0
    for r in range(rounds):
        outl.append(layers[off:off + (n >> r)])
        off += n >> r
    return outl, [roots[r].tobytes() for r in range(rounds)], [to_int(alphas[r]) for r in range(rounds - 1)], stream


def fri_sample_indices(seed: bytes, size: int, reduced_size: int, number: int):
    out = np.empty(number, dtype=np.uint64)
    s = np.frombuffer(seed, dtype=np.uint8).copy()
    lib.so_fri_sample_indices(p8(s), len(seed), size, reduced_size, number, p64(out))
    return [int(x) for x in out]

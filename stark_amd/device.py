"""Thin Python handle on the C ABI (include/stark_hip.h).

Host-path methods take numpy arrays of shape (n, 2) uint64 ({lo, hi} per field
element); *_dev methods take device pointers (e.g. torch tensors of dtype int64
and shape (n, 2), through .data_ptr()).  PyTorch is only used by callers for
device memory and streams; nothing here imports it."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib

ERR = {1: "ENOTPOW2", 2: "ETOOLARGE", 3: "ELENGTH", 4: "EINDEX", 5: "EROUNDS", 6: "EORDER", 7: "EARG", 8: "EHIP",
       9: "ENOMEM", 10: "ECALLBACK"}


class StarkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR.get(code, code)}: {msg}")
        self.code = code
        self.message = msg


def _fe(v: int):
    return (C.c_uint64 * 2)(v & 0xFFFFFFFFFFFFFFFF, v >> 64)


def _ptr(x) -> int:
    """numpy array -> host pointer; int -> as is; object with data_ptr() -> device pointer."""
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


def _as_fe_array(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim >= 1 and a.shape[-1] == 2, "field vectors are (..., 2) uint64"
    return a


class Tree:
    """A Merkle tree retained on the device (merkle.rs)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.h = ctx, handle
        ctx._track(self)

    @property
    def num_leaves(self) -> int:
        return self.ctx.lib.stark_tree_num_leaves(self.h)

    def open(self, index: int) -> List[bytes]:
        buf = (C.c_uint8 * (32 * 64))()
        depth = C.c_size_t(0)
        self.ctx._chk(self.ctx.lib.stark_merkle_open(self.h, index, buf, C.byref(depth)))
        raw = bytes(buf)
        return [raw[32 * i:32 * i + 32] for i in range(depth.value)]

    def open_many(self, indices: Sequence[int]) -> List[List[bytes]]:
        """open() for several indices in one launch."""
        idx = np.asarray(list(indices), dtype=np.uint64)
        out = np.empty((max(len(idx), 1), 64, 32), dtype=np.uint8)
        depth = C.c_size_t(0)
        self.ctx._chk(self.ctx.lib.stark_merkle_open_many(self.h, idx.ctypes.data, len(idx), out.ctypes.data,
                                                          C.byref(depth)))
        flat = out.reshape(-1, 32)
        d = depth.value
        return [[flat[i * d + k].tobytes() for k in range(d)] for i in range(len(idx))]

    def download(self) -> np.ndarray:
        n = self.num_leaves
        out = np.empty((2 * n - 1, 32), dtype=np.uint8)
        self.ctx._chk(self.ctx.lib.stark_tree_download(self.h, out.ctypes.data))
        return out

    def free(self):
        # a handle locks its context when it is freed: after Context.close() (which frees the handles
        # still alive) there is nothing left to do
        if self.h and self.ctx.h:
            self.ctx.lib.stark_tree_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FriLayers:
    """Codewords and trees of one Fri::commit, resident on the device."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.h = ctx, handle
        ctx._track(self)

    def __len__(self):
        return self.ctx.lib.stark_fri_layers_count(self.h)

    def layer_len(self, i: int) -> int:
        return self.ctx.lib.stark_fri_layer_len(self.h, i)

    def root(self, i: int) -> bytes:
        buf = (C.c_uint8 * 32)()
        self.ctx._chk(self.ctx.lib.stark_fri_layer_root(self.h, i, buf))
        return bytes(buf)

    def download(self, i: int) -> np.ndarray:
        out = np.empty((self.layer_len(i), 2), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_download(self.h, i, out.ctypes.data))
        return out

    def read(self, i: int, indices: Sequence[int]) -> np.ndarray:
        idx = np.asarray(indices, dtype=np.uint64)
        out = np.empty((len(idx), 2), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_read(
            self.h, i, idx.ctypes.data_as(_lib.u64p), len(idx), out.ctypes.data))
        return out

    def open(self, i: int, index: int) -> List[bytes]:
        buf = (C.c_uint8 * (32 * 64))()
        depth = C.c_size_t(0)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_open(self.h, i, index, buf, C.byref(depth)))
        raw = bytes(buf)
        return [raw[32 * k:32 * k + 32] for k in range(depth.value)]

    def open_many(self, i: int, indices: Sequence[int]) -> List[List[bytes]]:
        """open() for several indices of layer i in one launch."""
        idx = np.asarray(list(indices), dtype=np.uint64)
        out = np.empty((max(len(idx), 1), 64, 32), dtype=np.uint8)
        depth = C.c_size_t(0)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_open_many(self.h, i, idx.ctypes.data, len(idx), out.ctypes.data,
                                                             C.byref(depth)))
        flat = out.reshape(-1, 32)
        d = depth.value
        return [[flat[j * d + k].tobytes() for k in range(d)] for j in range(len(idx))]

    def query(self, i: int, c_indices: Sequence[int]):
        """One round of Fri::query on layers i and i+1: (values (3, count, 2) for a, b, c; paths_a, paths_b,
        paths_c as lists of digest lists), one round trip."""
        idx = np.asarray(list(c_indices), dtype=np.uint64)
        t = len(idx)
        vals = np.empty((3, max(t, 1), 2), dtype=np.uint64)
        paths = np.empty((max(t, 1) * 3 * 64, 32), dtype=np.uint8)
        d, dn = C.c_size_t(0), C.c_size_t(0)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_query(self.h, i, idx.ctypes.data, t, vals.ctypes.data,
                                                         paths.ctypes.data, C.byref(d), C.byref(dn)))
        d, dn = d.value, dn.value

        def cut(base, depth):
            return [[paths[base + j * depth + k].tobytes() for k in range(depth)] for j in range(t)]
        return vals[:, :t], cut(0, d), cut(t * d, d), cut(2 * t * d, dn)

    def query_all(self, top_indices: Sequence[int]):
        """Every round of Fri::prove's query phase in one round trip: a list, one entry per round, of what
        query(i, [t % (len_i / 2) for t in top_indices]) returns."""
        idx = np.asarray(list(top_indices), dtype=np.uint64)
        t, rounds = len(idx), len(self) - 1
        if t == 0 or rounds <= 0:
            return []
        depth = [int(self.layer_len(i)).bit_length() - 1 for i in range(rounds + 1)]
        digests = sum(t * (2 * depth[i] + depth[i + 1]) for i in range(rounds))
        vals = np.empty((rounds, 3, t, 2), dtype=np.uint64)
        paths = np.empty((max(digests, 1), 32), dtype=np.uint8)
        self.ctx._chk(self.ctx.lib.stark_fri_query_all(self.h, idx.ctypes.data, t, vals.ctypes.data, paths.ctypes.data))
        out, at = [], 0
        for i in range(rounds):
            d, dn = depth[i], depth[i + 1]

            def cut(base, dd):
                return [[paths[at + base + j * dd + k].tobytes() for k in range(dd)] for j in range(t)]
            out.append((vals[i], cut(0, d), cut(t * d, d), cut(2 * t * d, dn)))
            at += t * (2 * d + dn)
        return out

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.stark_fri_layers_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.stark_ctx_create(device, C.byref(h))
        if rc:
            raise StarkError(rc, self.lib.stark_last_error().decode())
        self.h = h
        self.device = device
        self._owned = True
        self._stream = "own"                                     # the stream calls are queued on, as last set here
        self._handles: "weakref.WeakSet" = weakref.WeakSet()     # trees and layers that belong to this context

    @classmethod
    def wrap(cls, handle, device: int = 0) -> "Context":
        """A Context over a stark_ctx* somebody else created and will destroy (the C++ mirror's process-wide
        one, hostlib.lib.sh_context()): calls made here and calls the mirror makes share one stream and one
        set of cached tables and blocks."""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        self.h = C.c_void_p(handle)
        self.device = device
        self._owned = False
        self._stream = None                                      # unknown: the first set_stream always goes through
        self._handles = weakref.WeakSet()
        return self

    def _chk(self, rc: int):
        if rc:
            raise StarkError(rc, self.lib.stark_last_error().decode())

    def _track(self, handle):
        self._handles.add(handle)

    def close(self):
        """Frees the trees and layers still alive on this context, then the context (their handles
        point into it, so none may outlive it)."""
        if self.h:
            for handle in list(self._handles):
                handle.free()
            if self._owned:
                self.lib.stark_ctx_destroy(self.h)
            self.h = None

    def trim(self):
        """Give cached device memory (scratch blocks, tables, planes) back to the runtime."""
        self._chk(self.lib.stark_ctx_trim(self.h))

    def set_stream(self, hip_stream: Optional[int]):
        """Launch on the caller's stream (None / 0 = HIP's default stream)."""
        self._chk(self.lib.stark_ctx_set_stream(self.h, hip_stream or None))
        self._stream = int(hip_stream or 0)

    def follow_stream(self, hip_stream: int):
        """set_stream unless the context is already on that stream (what it was last pointed at through this
        object; a context shared with other code is re-pointed every time)."""
        if not self._owned or self._stream != int(hip_stream or 0):
            self.set_stream(hip_stream)

    def reset_stream(self):
        self._chk(self.lib.stark_ctx_reset_stream(self.h))
        self._stream = "own"

    def copy_d2d(self, dst_ptr, src_ptr, nbytes: int):
        self._chk(self.lib.stark_copy_d2d(self.h, _ptr(dst_ptr), _ptr(src_ptr), nbytes))

    def synchronize(self):
        self._chk(self.lib.stark_ctx_synchronize(self.h))

    # ---- NTT ------------------------------------------------------------
    def ntt(self, data, inverse: bool = False) -> np.ndarray:
        a = _as_fe_array(data).copy()
        self._chk(self.lib.stark_ntt(self.h, a.ctypes.data, a.shape[0], int(inverse)))
        return a

    def ntt_batch(self, data, inverse: bool = False) -> np.ndarray:
        a = _as_fe_array(data).copy()
        assert a.ndim == 3
        self._chk(self.lib.stark_ntt_batch(self.h, a.ctypes.data, a.shape[1], a.shape[0], a.shape[1], int(inverse)))
        return a

    def ntt_dev(self, dptr, n: int, inverse: bool = False, n_cols: int = 1, col_stride: Optional[int] = None):
        self._chk(self.lib.stark_ntt_batch_dev(self.h, _ptr(dptr), n, n_cols, col_stride or n, int(inverse)))

    def coset_lde_batch(self, coeffs, n_out: int, offset: int) -> np.ndarray:
        a = _as_fe_array(coeffs)
        if a.ndim == 2:
            a = a[None]
        n_cols, n_coeffs = a.shape[0], a.shape[1]
        out = np.empty((n_cols, n_out, 2), dtype=np.uint64)
        self._chk(self.lib.stark_coset_lde_batch(self.h, a.ctypes.data, n_coeffs, n_coeffs, n_cols, n_out,
                                                 _fe(offset), out.ctypes.data, n_out))
        return out

    def coset_lde_batch_dev(self, coeffs_ptr, n_coeffs: int, coeff_stride: int, n_cols: int, n_out: int,
                            offset: int, out_ptr, out_stride: int):
        self._chk(self.lib.stark_coset_lde_batch_dev(self.h, _ptr(coeffs_ptr), n_coeffs, coeff_stride, n_cols,
                                                     n_out, _fe(offset), _ptr(out_ptr), out_stride))

    def fourstep_twiddle_dev(self, dptr, n_total: int, row_start: int, n_rows: int, n_cols: int,
                             inverse: bool = False):
        self._chk(self.lib.stark_fourstep_twiddle_dev(self.h, _ptr(dptr), n_total, row_start, n_rows, n_cols,
                                                      int(inverse)))

    def fourstep_pack_dev(self, src_ptr, n_rows: int, n_cols: int, world: int, dst_ptr, n_total: int = 0,
                          row_start: int = 0, twiddle: int = 0):
        """dst[d][c][r] = src[r][d*(n_cols/world) + c], times w^((row_start + r) * column) if twiddle (+1 / -1)."""
        self._chk(self.lib.stark_fourstep_pack_dev(self.h, _ptr(src_ptr), n_rows, n_cols, world, n_total, row_start,
                                                   twiddle, _ptr(dst_ptr)))

    def fourstep_unpack_dev(self, src_ptr, world: int, n_rows: int, run: int, dst_ptr):
        """dst[b][s][a] = src[s][b][a]."""
        self._chk(self.lib.stark_fourstep_unpack_dev(self.h, _ptr(src_ptr), world, n_rows, run, _ptr(dst_ptr)))

    # ---- polynomials as device vectors --------------------------------------
    def vec_mul_dev(self, a_ptr, b_ptr, out_ptr, n: int):
        self._chk(self.lib.stark_vec_mul_dev(self.h, _ptr(a_ptr), _ptr(b_ptr), _ptr(out_ptr), n))

    def vec_inv_dev(self, a_ptr, out_ptr, n: int):
        """out[i] = 1 / a[i] (0 for 0)."""
        self._chk(self.lib.stark_vec_inv_dev(self.h, _ptr(a_ptr), _ptr(out_ptr), n))

    def poly_compose_linear_dev(self, coeffs_ptr, n: int, s: int, n_cols: int = 1, col_stride: Optional[int] = None):
        """Coefficient j (lowest first) of every column times s^j."""
        self._chk(self.lib.stark_poly_compose_linear_dev(self.h, _ptr(coeffs_ptr), n, n_cols, col_stride or n, _fe(s)))

    def coset_interpolate_batch_dev(self, data_ptr, n: int, offset: int, n_cols: int = 1,
                                    col_stride: Optional[int] = None):
        """Values on offset * omega_n^i -> the n coefficients, in place."""
        self._chk(self.lib.stark_coset_interpolate_batch_dev(self.h, _ptr(data_ptr), n, n_cols, col_stride or n,
                                                             _fe(offset)))

    def poly_lincomb_dev(self, terms, out_ptr, n_out: int):
        """terms: (device pointer, length, shift, weight) each; out[j] = sum weight * src[j - shift]."""
        k = len(terms)
        srcs = (C.c_void_p * max(k, 1))(*[_ptr(t[0]) for t in terms])
        lens = (C.c_size_t * max(k, 1))(*[t[1] for t in terms])
        shifts = (C.c_size_t * max(k, 1))(*[t[2] for t in terms])
        w = (C.c_uint64 * max(2 * k, 2))()
        for i, t in enumerate(terms):
            w[2 * i], w[2 * i + 1] = t[3] & 0xFFFFFFFFFFFFFFFF, t[3] >> 64
        self._chk(self.lib.stark_poly_lincomb_dev(self.h, k, srcs, lens, shifts, w, _ptr(out_ptr), n_out))

    def mpoly_eval_dev(self, cols, rotations, terms, out_ptr, n: int, mul_ptr=None):
        """terms: (coefficient, exponent tuple over the columns) each; out[i] = sum c * prod cols[v][(i + rot[v]) % n]^e."""
        nv, k = len(cols), len(terms)
        ptrs = (C.c_void_p * nv)(*[_ptr(c) for c in cols])
        rot = (C.c_size_t * nv)(*(rotations or [0] * nv))
        w = (C.c_uint64 * max(2 * k, 2))()
        ex = (C.c_uint8 * max(k * nv, 1))()
        for i, (c, e) in enumerate(terms):
            w[2 * i], w[2 * i + 1] = c & 0xFFFFFFFFFFFFFFFF, c >> 64
            for v in range(nv):
                ex[i * nv + v] = e[v] if v < len(e) else 0
        self._chk(self.lib.stark_mpoly_eval_dev(self.h, nv, ptrs, rot, k, w, ex,
                                                None if mul_ptr is None else _ptr(mul_ptr), _ptr(out_ptr), n))

    def poly_degree_post_dev(self, coeffs_ptr, n: int, result_ptr):
        """poly_degree_dev without the wait: 0 for the zero polynomial, else degree + 1, as one uint64 at result_ptr."""
        self._chk(self.lib.stark_poly_degree_post_dev(self.h, _ptr(coeffs_ptr), n, _ptr(result_ptr)))

    def poly_degree_dev(self, coeffs_ptr, n: int):
        """(degree, is_zero) of n coefficients, lowest first."""
        d, z = C.c_size_t(0), C.c_int32(0)
        self._chk(self.lib.stark_poly_degree_dev(self.h, _ptr(coeffs_ptr), n, C.byref(d), C.byref(z)))
        return d.value, bool(z.value)

    def sample_bytes_dev(self, bytes_ptr, width: int, n: int, out_ptr, reversed_: bool = False):
        self._chk(self.lib.stark_sample_bytes_dev(self.h, _ptr(bytes_ptr), width, n, int(reversed_), _ptr(out_ptr)))

    def poly_eval_points_dev(self, coeffs_ptr, n: int, points: Sequence[int], n_polys: int = 1,
                             stride: Optional[int] = None) -> List[List[int]]:
        """Polynomial::eval of n_polys resident polynomials (n coefficients each, lowest first) at up to 16
        points: result[q][p]."""
        pts = np.array([[x & ((1 << 64) - 1), x >> 64] for x in points], dtype=np.uint64).reshape(-1, 2)
        out = np.zeros((max(n_polys * len(points), 1), 2), dtype=np.uint64)
        self._chk(self.lib.stark_poly_eval_points_dev(self.h, _ptr(coeffs_ptr), n, n_polys, stride or n, pts.ctypes.data,
                                                      len(points), out.ctypes.data))
        vals = [int(out[i, 0]) | (int(out[i, 1]) << 64) for i in range(n_polys * len(points))]
        return [vals[q * len(points):(q + 1) * len(points)] for q in range(n_polys)]

    def vec_mul(self, a, b) -> np.ndarray:
        a, b = _as_fe_array(a), _as_fe_array(b)
        assert a.shape == b.shape
        out = np.empty_like(a)
        self._chk(self.lib.stark_vec_mul(self.h, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0]))
        return out

    # ---- Merkle -----------------------------------------------------------
    def merkle_commit_fe(self, elems, keep_tree: bool = False):
        a = _as_fe_array(elems)
        root = (C.c_uint8 * 32)()
        th = C.c_void_p()
        self._chk(self.lib.stark_merkle_commit_fe(self.h, a.ctypes.data, a.shape[0], root,
                                                  C.byref(th) if keep_tree else None))
        return (bytes(root), Tree(self, th)) if keep_tree else bytes(root)

    def merkle_commit_fe_dev(self, dptr, n: int, keep_tree: bool = False):
        root = (C.c_uint8 * 32)()
        th = C.c_void_p()
        self._chk(self.lib.stark_merkle_commit_fe_dev(self.h, _ptr(dptr), n, root,
                                                      C.byref(th) if keep_tree else None))
        return (bytes(root), Tree(self, th)) if keep_tree else bytes(root)

    def merkle_roots_fe_batch_dev(self, dptr, n: int, n_cols: int, col_stride: Optional[int] = None) -> List[bytes]:
        """One Merkle::commit per column (column c = n elements at dptr + c*col_stride), all columns
        through the same launches; returns the n_cols roots."""
        roots = (C.c_uint8 * (32 * max(n_cols, 1)))()
        self._chk(self.lib.stark_merkle_roots_fe_batch_dev(self.h, _ptr(dptr), n, n_cols,
                                                           n if col_stride is None else col_stride, roots))
        raw = bytes(roots)
        return [raw[32 * c:32 * (c + 1)] for c in range(n_cols)]

    def merkle_root_fe_static_dev(self, dptr, n: int, root_dptr):
        """Root of the n leaves at dptr into 32 bytes of device memory at root_dptr; the tree lives in the
        library's static device array (n <= 2^18), nothing is allocated and nothing is synchronised."""
        self._chk(self.lib.stark_merkle_root_fe_static_dev(self.h, _ptr(dptr), n, _ptr(root_dptr)))

    def merkle_subtree_roots_dev(self, dptr, chunk_len: int, n_chunks: int, roots_dptr, keep_forest: bool = False):
        """Subtree root of each of n_chunks aligned runs of chunk_len leaves, left in device memory.
        With keep_forest returns a Tree whose open(local leaf index) gives the siblings inside the subtree."""
        th = C.c_void_p()
        self._chk(self.lib.stark_merkle_subtree_roots_dev(self.h, _ptr(dptr), chunk_len, n_chunks, _ptr(roots_dptr),
                                                          C.byref(th) if keep_forest else None))
        return Tree(self, th) if keep_forest else None

    def merkle_root_of_digests_dev(self, dptr, m: int, keep_tree: bool = False):
        """Root of the tree whose lowest level is the m digests at dptr (device memory)."""
        root = (C.c_uint8 * 32)()
        th = C.c_void_p()
        self._chk(self.lib.stark_merkle_root_of_digests_dev(self.h, _ptr(dptr), m, root,
                                                            C.byref(th) if keep_tree else None))
        return (bytes(root), Tree(self, th)) if keep_tree else bytes(root)

    def merkle_commit_bytes(self, leaves: Sequence[bytes], keep_tree: bool = False):
        flat = np.frombuffer(b"".join(leaves) or b"\0", dtype=np.uint8).copy()
        offs = np.zeros(len(leaves) + 1, dtype=np.uint64)
        if leaves:
            offs[1:] = np.cumsum([len(l) for l in leaves])
        root = (C.c_uint8 * 32)()
        th = C.c_void_p()
        self._chk(self.lib.stark_merkle_commit_bytes(self.h, flat.ctypes.data, offs.ctypes.data, len(leaves), root,
                                                     C.byref(th) if keep_tree else None))
        return (bytes(root), Tree(self, th)) if keep_tree else bytes(root)

    # ---- FRI --------------------------------------------------------------
    def fri_fold(self, codeword, alpha: int, offset: int, omega: int) -> np.ndarray:
        a = _as_fe_array(codeword)
        out = np.empty((a.shape[0] // 2, 2), dtype=np.uint64)
        self._chk(self.lib.stark_fri_fold(self.h, a.ctypes.data, a.shape[0], _fe(alpha), _fe(offset), _fe(omega),
                                          out.ctypes.data))
        return out

    def fri_fold_dev(self, cw_ptr, n: int, alpha: int, offset: int, omega: int, out_ptr):
        self._chk(self.lib.stark_fri_fold_dev(self.h, _ptr(cw_ptr), n, _fe(alpha), _fe(offset), _fe(omega),
                                              _ptr(out_ptr)))

    def fri_fold_shard_dev(self, local_ptr, local_len: int, alpha: int, offset: int, omega: int, chunk_len: int,
                           period: int, base: int, out_ptr, chunk_roots_ptr=None, keep_forest: bool = False):
        th = C.c_void_p()
        self._chk(self.lib.stark_fri_fold_shard_dev(self.h, _ptr(local_ptr), local_len, _fe(alpha), _fe(offset),
                                                    _fe(omega), chunk_len, period, base, _ptr(out_ptr),
                                                    None if chunk_roots_ptr is None else _ptr(chunk_roots_ptr),
                                                    C.byref(th) if keep_forest else None))
        return Tree(self, th) if keep_forest else None

    def fri_num_rounds(self, domain_length: int, expansion_factor: int, num_colinearity_tests: int) -> int:
        return self.lib.stark_fri_num_rounds(domain_length, expansion_factor, num_colinearity_tests)

    def fri_commit(self, codeword, offset: int, omega: int, expansion_factor: int, num_colinearity_tests: int,
                   transcript: Callable[[int, bool, bytes], Optional[int]], n: Optional[int] = None) -> FriLayers:
        """transcript(round, last, root) -> alpha (int) for every round but the last.
        `codeword` is a numpy array (host path) or a device pointer / tensor (then pass n)."""
        err: List[BaseException] = []

        def _cb(_user, rnd, last, root_p, alpha_out):
            try:
                root = C.string_at(root_p, 32)
                alpha = transcript(int(rnd), bool(last), root)
                if not last:
                    alpha_out[0] = alpha & 0xFFFFFFFFFFFFFFFF
                    alpha_out[1] = alpha >> 64
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through C
                err.append(e)
                return 1

        cb = _lib.TRANSCRIPT_CB(_cb)
        lh = C.c_void_p()
        if isinstance(codeword, np.ndarray):
            a = _as_fe_array(codeword)
            rc = self.lib.stark_fri_commit(self.h, a.ctypes.data, a.shape[0], _fe(offset), _fe(omega),
                                           expansion_factor, num_colinearity_tests, cb, None, C.byref(lh))
        else:
            rc = self.lib.stark_fri_commit_dev(self.h, _ptr(codeword), n, _fe(offset), _fe(omega),
                                               expansion_factor, num_colinearity_tests, cb, None, C.byref(lh))
        if err:
            raise err[0]
        self._chk(rc)
        return FriLayers(self, lh)

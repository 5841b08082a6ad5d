"""Python handle on the long-trace prover of the C++ mirror (stark_amd/host/stark_scaled.hpp through the shims
of host_capi.cpp): a chain of Rescue-Prime rounds as the statement, proofs on one GPU, and the prover's steps
one at a time for stark_amd.sharded_stark.stark_prove_sharded."""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import hostlib
from .device import Context

_lib = hostlib.lib
_lib.sh_chain_new.restype = C.c_void_p
_lib.sh_chain_random_bytes.restype = C.c_size_t
_lib.sh_chain_fri_domain_length.restype = C.c_size_t
_lib.sh_session_begin.restype = C.c_void_p
_lib.sh_session_coefficients.restype = C.c_void_p

GENERATOR = 85408008396924667383611388730472331217
TIMING_NAMES = ("host", "interpolate", "quotients", "lde", "merkle", "fri", "openings", "total")
P = 407 * (1 << 119) + 1


def mirror_context() -> Context:
    """The process-wide device context the mirror runs on, as a Context (not owned)."""
    h = _lib.sh_context()
    if not h:
        raise hostlib.HostPanic(_lib.sh_last_panic().decode())
    return Context.wrap(h)


class ChainJob:
    """A Rescue-Prime chain of `rows` rows (the statement: output = the chain's last state) with its trace, AIR,
    constant columns and the ScaledStark sized for it."""

    def __init__(self, input_element: int, rows: int, expansion: int = 4, num_colinearity_tests: int = 2,
                 security: int = 2):
        self.h = C.c_void_p(_lib.sh_chain_new(hostlib.fe(input_element), C.c_size_t(rows), C.c_size_t(expansion),
                                              C.c_size_t(num_colinearity_tests), C.c_size_t(security)))
        if not self.h:
            raise hostlib.HostPanic(_lib.sh_last_panic().decode())
        self.rows, self.expansion, self.num_colinearity_tests = rows, expansion, num_colinearity_tests

    def free(self):
        if self.h:
            _lib.sh_chain_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def random_bytes(self) -> int:
        """Random bytes one proof draws."""
        return _lib.sh_chain_random_bytes(self.h)

    @property
    def fri_domain_length(self) -> int:
        return _lib.sh_chain_fri_domain_length(self.h)

    def output(self) -> int:
        out = (C.c_uint64 * 2)()
        hostlib.chk(_lib.sh_chain_output(self.h, out))
        return hostlib.val(out)

    def prove(self, random: bytes, by_columns: bool = False, proof_capacity: int = 1 << 22):
        """(proof bytes, timings in ms by name).  by_columns hands the trace over column by column
        (ScaledStark::prove_columns) instead of row by row."""
        rbuf = (C.c_uint8 * len(random)).from_buffer_copy(random)
        proof = (C.c_uint8 * proof_capacity)()
        plen = C.c_size_t(proof_capacity)
        tm = (C.c_double * 8)()
        fn = _lib.sh_chain_prove_columns if by_columns else _lib.sh_chain_prove
        hostlib.chk(fn(self.h, rbuf, C.c_size_t(len(random)), proof, C.byref(plen), tm))
        return bytes(proof[:plen.value]), dict(zip(TIMING_NAMES, tm))

    def verify(self, proof: bytes, output: Optional[int] = None) -> bool:
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        rc = _lib.sh_chain_verify(self.h, hostlib.fe(self.output() if output is None else output), buf,
                                  C.c_size_t(len(proof)))
        return bool(hostlib.chk(rc))


class ChainSteps:
    """The steps of ScaledStark::prove off the FRI domain, one at a time, on this process's GPU; the interface
    stark_prove_sharded asks of `steps`.  Results come back as torch tensors (copies on `ctx`'s stream, which is
    the mirror's own context, so the copies are ordered behind the step that produced them)."""

    def __init__(self, job: ChainJob, random: bytes, ctx: Context):
        import torch
        self.torch = torch
        self.job, self.ctx = job, ctx
        self._random = (C.c_uint8 * len(random)).from_buffer_copy(random)      # read until the session is freed
        self.h = C.c_void_p(_lib.sh_session_begin(job.h, self._random, C.c_size_t(len(random))))
        if not self.h:
            raise hostlib.HostPanic(_lib.sh_last_panic().decode())
        shape = (C.c_uint64 * 8)()
        _lib.sh_session_shape(self.h, shape)
        (self.registers, self.bq_len, self.bq_stride, self.poly_len, self.fri_domain_length, self.expansion_factor,
         self.num_colinearity_tests, _) = [int(x) for x in shape]
        self.offset = GENERATOR                                                  # FieldElement::generator (field.rs:29)
        self.omega = hostlib.primitive_nth_root(self.fri_domain_length)
        self.timings = None

    def _copy(self, which: int, cols: int, length: int, stride: int):
        torch = self.torch
        src = _lib.sh_session_coefficients(self.h, which)
        if not src:
            raise RuntimeError("step result not available yet")
        out = torch.empty((cols, length, 2), dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
        self.ctx.follow_stream(torch.cuda.current_stream().cuda_stream)
        for c in range(cols):
            self.ctx.copy_d2d(out[c].data_ptr(), src + 16 * c * stride, 16 * length)
        return out

    def boundary_quotients(self):
        return self._copy(0, self.registers, self.bq_len, self.bq_stride)

    def randomizer(self):
        hostlib.chk(_lib.sh_session_transition(self.h))
        return self._copy(1, 1, self.poly_len, self.poly_len)

    def combination(self, seed: bytes):
        buf = (C.c_uint8 * len(seed)).from_buffer_copy(seed)
        hostlib.chk(_lib.sh_session_combine(self.h, buf, C.c_size_t(len(seed))))
        return self._copy(2, 1, self.poly_len, self.poly_len)

    def finish(self):
        tm = (C.c_double * 8)()
        hostlib.chk(_lib.sh_session_finish(self.h, tm))
        self.timings = dict(zip(TIMING_NAMES, tm))

    def free(self):
        if self.h:
            _lib.sh_session_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

"""Stark::prove / Stark::verify of the C++ mirror (stark_amd/host/stark_prover.hpp) with every
low-degree extension, commitment, FRI layer and opening on the device, against the Python
restatement fed the same random bytes: the two proofs must be the same bytes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import stark_oracle as so
import stark_prover_oracle as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hm():
    import host_mirror
    return host_mirror


def _random_buffer(seed, draws):
    rng = sp.counter_rng(seed)
    return b"".join(rng(17) for _ in range(draws))


def _prove(hm, inp, seed, expansion=4, ncol=2, security=2):
    buf = _random_buffer(seed, 256)
    rb = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
    proof = (C.c_uint8 * (1 << 20))()
    plen = C.c_size_t(len(proof))
    out = (C.c_uint64 * 2)()
    used = hm.chk(hm.lib.sh_stark_rescue_prove(hm.fe(inp), C.c_size_t(expansion), C.c_size_t(ncol), C.c_size_t(security),
                                               rb, C.c_size_t(len(buf)), proof, C.byref(plen), out))
    return bytes(proof[:plen.value]), hm.val(out), used


def _verify(hm, out, proof, expansion=4, ncol=2, security=2):
    pb = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
    return hm.lib.sh_stark_rescue_verify(hm.fe(out), C.c_size_t(expansion), C.c_size_t(ncol), C.c_size_t(security), pb,
                                         C.c_size_t(len(proof)))


@pytest.mark.parametrize("seed,inp", [
    (b"stark-test", so.fe_sample(bytes([48, 120, 100, 101, 97, 100, 98, 101, 101, 102]))),     # stark.rs:740
    (b"second", 1),
])
def test_proof_bytes_equal_restatement(hm, seed, inp):
    rp = sp.RescuePrime()
    st = sp.Stark(4, 2, 2, rp.m, rp.N + 1)
    air = rp.transition_constraints(st.omicron)
    out = rp.hash(inp)
    boundary = rp.boundary_constraints(out)
    want = st.prove(rp.trace(inp), air, boundary, sp.counter_rng(seed))
    got, got_out, used = _prove(hm, inp, seed)
    assert got_out == out
    assert used == 17 * (8 * 2 + 128)                  # 8 randomizer rows x 2 registers + 128 randomizer coefficients
    assert got == want
    assert _verify(hm, out, got) == 1                  # the mirror's verifier
    assert st.verify(got, air, boundary)               # the restatement's verifier


def test_verifier_rejects(hm):
    inp = 12345
    proof, out, _ = _prove(hm, inp, b"reject")
    assert _verify(hm, out, proof) == 1
    assert _verify(hm, so.fe_add(out, 1), proof) == 0              # wrong claimed output
    ps = so.ProofStream.deserialize(proof)
    objs = list(ps.objects)
    # swap one opened boundary-quotient leaf for another value: its authentication path no longer fits
    k = len(objs) - 2 * 4 - 2
    objs[k] = so.json_field_element(7)
    forged = so.ProofStream()
    for o in objs:
        forged.push(o)
    assert _verify(hm, out, forged.serialize()) == 0
    # a proof for another input does not verify against this output
    other, other_out, _ = _prove(hm, inp + 1, b"reject")
    assert other_out != out and _verify(hm, out, other) == 0


def test_prover_needs_its_random_bytes(hm):
    rb = (C.c_uint8 * 16)()
    plen = C.c_size_t(0)
    rc = hm.lib.sh_stark_rescue_prove(hm.fe(3), C.c_size_t(4), C.c_size_t(2), C.c_size_t(2), rb, C.c_size_t(16), None,
                                      C.byref(plen), None)
    assert rc < 0 and b"random bytes exhausted" in hm.lib.sh_last_panic()

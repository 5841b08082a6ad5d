"""The C++ host mirror of the reference API (stark_amd/host/stark_host.hpp):
everything that runs without a GPU -- scalars, encodings, hashes, transcript,
index sampling, Merkle::verify and the whole of Fri::verify -- against the
Python restatement.  Reads like the reference's own unit tests."""
import ctypes as C
import hashlib
import random
import subprocess
import os

import numpy as np
import pytest

import stark_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, G = so.P, so.GENERATOR


@pytest.fixture(scope="module")
def hm():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stark_amd", "host")])
    import host_mirror
    return host_mirror


def test_field_ops(hm):
    """field.rs:162-276."""
    random.seed(7)
    vals = [random.randrange(P) for _ in range(200)] + [0, 1, 2, P - 1, P - 2, 1 << 64, (1 << 64) - 1]
    for _ in range(300):
        a, b = random.choice(vals), random.choice(vals)
        assert hm.fe_op(0, a, b) == so.fe_add(a, b)
        assert hm.fe_op(1, a, b) == so.fe_sub(a, b)
        assert hm.fe_op(2, a, b) == so.fe_mul(a, b)
        assert hm.fe_op(4, a) == so.fe_neg(a)
    for a in vals[:40] + [0, 1, 3]:
        assert hm.fe_op(5, a) == so.fe_inverse(a)
        if a:
            assert hm.fe_op(2, a, hm.fe_op(5, a)) == 1
            assert hm.fe_op(3, 5, a) == so.fe_div(5, a)
    assert hm.fe_op(0, P - 1, 10) == 9                        # test_addition
    assert hm.fe_op(1, 2, 3) == P - 1                         # test_negative_handling
    out = (C.c_uint64 * 2)()
    hm.chk(hm.lib.sh_fe_pow(hm.fe(2), hm.fe(32), out))
    assert hm.val(out) == 2**32                               # test_pow
    hm.chk(hm.lib.sh_fe_pow(hm.fe(G), hm.fe(1 << 119), out))
    assert hm.val(out) == 1


def test_primitive_root_and_sample(hm, ref_kats, golden):
    out = (C.c_uint64 * 2)()
    hm.chk(hm.lib.sh_primitive_nth_root(hm.fe(128), out))
    assert hm.val(out) == int(ref_kats["omicron_128th_root"])          # stark.rs:65
    for k, v in golden["roots"].items():
        hm.chk(hm.lib.sh_primitive_nth_root(hm.fe(1 << int(k)), out))
        assert hm.val(out) == int(v)
    with pytest.raises(hm.HostPanic, match="nth root of unity"):
        hm.chk(hm.lib.sh_primitive_nth_root(hm.fe(3), out))
    for hx, v in golden["sample"]:
        b = bytes.fromhex(hx)
        hm.chk(hm.lib.sh_fe_sample(b, C.c_size_t(len(b)), out))
        assert hm.val(out) == int(v)


def test_display_and_bincode(hm, golden):
    buf = C.create_string_buffer(64)
    for v in [0, 1, 10, P - 1, 2**64, 123456789012345678901234567890]:
        n = hm.lib.sh_fe_to_string(hm.fe(v), buf)
        assert buf.raw[:n].decode() == str(v)
    b = (C.c_uint8 * 25)()
    for v, hx in golden["leaf_bytes"]:
        n = hm.lib.sh_bincode_fe(hm.fe(int(v)), b)
        assert bytes(b[:n]).hex() == hx


@pytest.mark.parametrize("n", [0, 1, 64, 127, 128, 129, 255, 256, 257, 1000])
def test_hashes_vs_hashlib(hm, n):
    data = hashlib.shake_256(bytes([n % 251])).digest(n)
    for outlen in (32, 64):
        out = (C.c_uint8 * outlen)()
        hm.lib.sh_blake2b(data, C.c_size_t(n), out, C.c_size_t(outlen))
        assert bytes(out) == hashlib.blake2b(data, digest_size=outlen).digest()
    for outlen in (32, 200):
        out = (C.c_uint8 * outlen)()
        hm.lib.sh_shake256(data, C.c_size_t(n), out, C.c_size_t(outlen))
        assert bytes(out) == hashlib.shake_256(data).digest(outlen)


def test_proof_stream(hm):
    """proof_stream.rs:76-95."""
    p1 = hm.Stream.from_objects(["1", "2", "3", "4"])
    ref = so.ProofStream()
    for s in ("1", "2", "3", "4"):
        ref.push(s)
    assert p1.serialize() == ref.serialize()
    p2 = hm.Stream.deserialize(p1.serialize())
    assert p2.objects == ["1", "2", "3", "4"]
    assert p1.fiat_shamir(32) == p2.fiat_shamir(32) == ref.prover_fiat_shamir(32)
    assert p2.fiat_shamir(32, verifier=True) == hashlib.shake_256(so.bincode_vec_string([])).digest(32)
    n = C.c_size_t()
    for _ in range(4):
        hm.chk(hm.lib.sh_ps_pull(p2.h, None, C.byref(n)))
    assert p2.fiat_shamir(32, verifier=True) == ref.prover_fiat_shamir(32)
    with pytest.raises(hm.HostPanic, match="queue empty"):
        hm.chk(hm.lib.sh_ps_pull(p2.h, None, C.byref(n)))


def test_sample_indices(hm):
    fri = so.Fri(G, so.primitive_nth_root(256), 256, 4, 17)
    for seed, size, red, num in ((b"\x01" * 32, 128, 128, 17), (hashlib.sha256(b"x").digest(), 1 << 23, 128, 17),
                                 (b"", 64, 16, 8)):
        out = np.zeros(num, dtype=np.uint64)
        hm.chk(hm.lib.sh_fri_sample_indices(seed, C.c_size_t(len(seed)), C.c_size_t(size), C.c_size_t(red),
                                            C.c_size_t(num), hm.p(out)))
        assert [int(x) for x in out] == fri.sample_indices(seed, size, red, num)


def test_merkle_verify_and_host_commit(hm):
    """merkle.rs:157-206 (verification half; commit/open run on the device)."""
    n = 64
    leaves = [hashlib.shake_256(bytes([i])).digest(1 + (i * 7) % 99) for i in range(n)]
    flat, offs = hm.flatten(leaves)
    root = (C.c_uint8 * 32)()
    hm.chk(hm.lib.sh_merkle_commit_host(hm.p(flat), hm.p(offs), C.c_size_t(n), root))
    assert bytes(root) == so.merkle_commit(leaves)
    for i in (0, 1, 31, 63):
        path = so.merkle_open(i, leaves)
        pb = b"".join(path)
        ok = hm.chk(hm.lib.sh_merkle_verify(bytes(root), C.c_size_t(i), pb, C.c_size_t(len(path)), leaves[i],
                                            C.c_size_t(len(leaves[i]))))
        assert ok == 1
        assert hm.chk(hm.lib.sh_merkle_verify(bytes(root), C.c_size_t((i + 1) % n), pb, C.c_size_t(len(path)),
                                              leaves[i], C.c_size_t(len(leaves[i])))) == 0
        bad = bytes(32) + pb[32:]
        assert hm.chk(hm.lib.sh_merkle_verify(bytes(root), C.c_size_t(i), bad, C.c_size_t(len(path)), leaves[i],
                                              C.c_size_t(len(leaves[i])))) == 0
    with pytest.raises(hm.HostPanic, match="invalid index"):
        hm.chk(hm.lib.sh_merkle_verify(bytes(root), C.c_size_t(64), b"".join(so.merkle_open(0, leaves)), C.c_size_t(6),
                                       leaves[0], C.c_size_t(len(leaves[0]))))
    with pytest.raises(hm.HostPanic, match="power of two"):
        hm.chk(hm.lib.sh_merkle_commit_host(hm.p(flat), hm.p(offs), C.c_size_t(3), root))


def test_constructor_panics_match_reference_texts(hm):
    d = np.zeros((12, 2), dtype=np.uint64)
    with pytest.raises(hm.HostPanic, match="NTT size must be power of 2"):
        hm.chk(hm.lib.sh_ntt(hm.p(d), C.c_size_t(12), C.c_size_t(12), 0))
    with pytest.raises(hm.HostPanic, match="Cannot do FRI with less than 1 round"):
        hm.chk(hm.lib.sh_fri_num_rounds(hm.fe(G), hm.fe(1), C.c_size_t(4), C.c_size_t(4), C.c_size_t(1)))
    assert hm.chk(hm.lib.sh_fri_num_rounds(hm.fe(G), hm.fe(1), C.c_size_t(1 << 24), C.c_size_t(4), C.c_size_t(17))) == 18


def test_fri_verify_accepts_oracle_proof_and_rejects_corruption(hm):
    """fri.rs:428-504 with the prover replaced by the Python restatement: the
    C++ verifier must accept its proof, return the opened points, and reject
    the corrupted codeword's proof."""
    degree, expansion, ncol = 63, 4, 17
    n0 = (degree + 1) * expansion
    omega = so.primitive_nth_root(n0)
    fri = so.Fri(G, omega, n0, expansion, ncol)
    coeffs = [so.fe_new(i) for i in range(degree + 1)]
    codeword = so.poly_eval_domain(coeffs, [so.fe_pow(omega, i) for i in range(n0)])
    ps = so.ProofStream()
    fri.prove(codeword, ps)
    ok, points = hm.fri_verify(G, omega, n0, expansion, ncol, hm.Stream.from_objects(ps.objects))
    assert ok and len(points) == 2 * ncol
    for idx, v in points:
        assert so.poly_eval(coeffs, so.fe_pow(omega, idx)) == v
    ref_points = []
    ps.read_idx = 0
    assert so.Fri(G, omega, n0, expansion, ncol).verify(ps, ref_points)
    assert points == ref_points
    bad = list(codeword)
    for i in range(degree // 3):
        bad[i] = 0
    ps = so.ProofStream()
    fri.prove(bad, ps)
    ok, _ = hm.fri_verify(G, omega, n0, expansion, ncol, hm.Stream.from_objects(ps.objects))
    assert ok is False
    # a tampered authentication path is rejected
    ps = so.ProofStream()
    fri.prove(codeword, ps)
    objs = list(ps.objects)
    import json
    path = json.loads(objs[-1])
    path[0][0] ^= 1
    objs[-1] = json.dumps(path, separators=(",", ":"))
    ok, _ = hm.fri_verify(G, omega, n0, expansion, ncol, hm.Stream.from_objects(objs))
    assert ok is False

"""The C++ host mirror driving the GPU through the C ABI: NTT / Polynomial /
Merkle / Fri with the reference's signatures, checked against the Python
restatement.  Fri::prove must give the restatement's proof stream byte for
byte, and each side's verifier must accept the other's proof."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_c as oc          # noqa: E402
import stark_oracle as so      # noqa: E402
import host_mirror as hm       # noqa: E402

P, G = so.P, so.GENERATOR
sz = C.c_size_t


def _ntt(vals, n, inverse=False):
    a = hm.arr(vals)
    hm.chk(hm.lib.sh_ntt(hm.p(a), sz(len(vals)), sz(n), int(inverse)))
    return hm.ints(a)


def test_ntt_reference_unit_tests():
    """ntt.rs:143-206."""
    coeffs = [1, 2, 3, 4]
    out = np.zeros((4, 2), dtype=np.uint64)
    hm.chk(hm.lib.sh_ntt_evaluate(hm.p(hm.arr(coeffs)), sz(4), sz(4), hm.p(out)))
    assert _ntt(hm.ints(out), 4, inverse=True) == coeffs
    evals = [i * i + 1 for i in range(8)]
    c = _ntt(evals, 8, inverse=True)
    assert _ntt(c, 8) == evals
    prod = np.zeros((3, 2), dtype=np.uint64)
    hm.chk(hm.lib.sh_ntt_multiply(hm.p(hm.arr([1, 1])), sz(2), hm.p(hm.arr([1, 2])), sz(2), hm.p(prod)))
    assert hm.ints(prod) == [1, 3, 2]
    with pytest.raises(hm.HostPanic, match="Input size must match NTT size"):
        _ntt([1, 2, 3], 8)


@pytest.mark.parametrize("n", [8, 256, 8192])
def test_ntt_vs_restatement(n):
    x = so.splitmix_field_vector(n, seed=n)
    f = list(x)
    so.NTT(n).forward(f)
    assert _ntt(x, n) == f
    assert _ntt(f, n, inverse=True) == x


def test_evaluate_truncates_like_resize():
    """ntt.rs:104: more coefficients than points are dropped."""
    coeffs = so.splitmix_field_vector(20, seed=3)
    out = np.zeros((16, 2), dtype=np.uint64)
    hm.chk(hm.lib.sh_ntt_evaluate(hm.p(hm.arr(coeffs)), sz(20), sz(16), hm.p(out)))
    assert hm.ints(out) == so.NTT(16).evaluate(coeffs)


def _eval_domain(coeffs_high_first, domain):
    out = np.zeros((len(domain), 2), dtype=np.uint64)
    hm.chk(hm.lib.sh_eval_domain(hm.p(hm.arr(coeffs_high_first)), sz(len(coeffs_high_first)), hm.p(hm.arr(domain)),
                                 sz(len(domain)), hm.p(out)))
    return hm.ints(out)


def test_eval_domain_all_branches():
    """univariate_poly.rs:44-54 and its tests :820-868."""
    for n in (16, 32, 1024):
        w = so.primitive_nth_root(n)
        dom = [so.fe_pow(w, i) for i in range(n)]
        coeffs = so.splitmix_field_vector(n, seed=n + 1)
        assert _eval_domain(coeffs, dom) == so.poly_eval_domain(coeffs, dom)
        # NTT path truncates the top coefficient (size+1 coefficients on size points)
        longer = so.splitmix_field_vector(n + 1, seed=n + 2)
        assert _eval_domain(longer, dom) == so.poly_eval_domain(longer, dom) == \
            [so.poly_eval(longer[1:], x) for x in dom]
        # shifted domain: the reference evaluates by Horner; the coset transform gives the same values
        shifted = [so.fe_mul(G, x) for x in dom]
        assert _eval_domain(coeffs, shifted) == [so.poly_eval(coeffs, x) for x in shifted]
        # shifted domain with too many coefficients: Horner (no truncation)
        assert _eval_domain(longer, shifted) == [so.poly_eval(longer, x) for x in shifted]
    # small and irregular domains: Horner
    coeffs = [5, 4, 3, 2, 1]
    for dom in ([1, 2, 3], [so.fe_pow(so.primitive_nth_root(4), i) for i in range(4)], [7, 7, 0, P - 1, 3, 9, 2, 11]):
        assert _eval_domain(coeffs, dom) == [so.poly_eval(coeffs, x) for x in dom]


def test_lagrange_paths():
    """univariate_poly.rs:127-177."""
    n = 32
    w = so.primitive_nth_root(n)
    x = [so.fe_pow(w, i) for i in range(n)]
    y = so.splitmix_field_vector(n, seed=5)
    out = np.zeros((64, 2), dtype=np.uint64)
    ln = sz()
    hm.chk(hm.lib.sh_lagrange(hm.p(hm.arr(x)), hm.p(hm.arr(y)), sz(n), hm.p(out), C.byref(ln)))
    assert ln.value == n and hm.ints(out[:n]) == so.lagrange(x, y) == so.lagrange_traditional(x, y)
    x3, y3 = [0, 1, 2], [2, 5, 5]                      # multivariate_poly.rs:277-296
    hm.chk(hm.lib.sh_lagrange(hm.p(hm.arr(x3)), hm.p(hm.arr(y3)), sz(3), hm.p(out), C.byref(ln)))
    assert hm.ints(out[:ln.value]) == so.lagrange(x3, y3)


def test_merkle_reference_unit_test():
    """merkle.rs:157-206 through Merkle::commit / open (device) and verify (host)."""
    n = 64
    leaves = [hashlib.shake_256(bytes([i])).digest(1 + (i * 7) % 99) for i in range(n)]
    flat, offs = hm.flatten(leaves)
    root = (C.c_uint8 * 32)()
    hm.chk(hm.lib.sh_merkle_commit(hm.p(flat), hm.p(offs), sz(n), root))
    assert bytes(root) == so.merkle_commit(leaves)
    path = (C.c_uint8 * (32 * 64))()
    depth = sz()
    for i in range(n):
        hm.chk(hm.lib.sh_merkle_open(sz(i), hm.p(flat), hm.p(offs), sz(n), path, C.byref(depth)))
        pb = bytes(path[: 32 * depth.value])
        assert [pb[32 * k:32 * k + 32] for k in range(depth.value)] == so.merkle_open(i, leaves)
        assert hm.chk(hm.lib.sh_merkle_verify(bytes(root), sz(i), pb, depth, leaves[i], sz(len(leaves[i])))) == 1
    with pytest.raises(hm.HostPanic, match="power of two"):
        hm.chk(hm.lib.sh_merkle_commit(hm.p(flat), hm.p(offs), sz(3), root))
    with pytest.raises(hm.HostPanic, match="cannot open invalid index"):
        hm.chk(hm.lib.sh_merkle_open(sz(64), hm.p(flat), hm.p(offs), sz(n), path, C.byref(depth)))


def _prove(cw_ints, offset, omega, expansion, ncol, by_value=False):
    n = len(cw_ints)
    stream = hm.Stream()
    cw = hm.arr(cw_ints)
    if by_value:
        hm.chk(hm.lib.sh_fri_prove_by_value(hm.p(cw), sz(n), hm.fe(offset), hm.fe(omega), sz(expansion), sz(ncol),
                                            stream.h))
        return stream, None
    idx = np.zeros(2 * ncol, dtype=np.uint64)
    k = hm.chk(hm.lib.sh_fri_prove(hm.p(cw), sz(n), hm.fe(offset), hm.fe(omega), sz(expansion), sz(ncol), stream.h,
                                   hm.p(idx)))
    return stream, [int(i) for i in idx[:k]]


def test_fri_reference_unit_test(golden):
    """fri.rs:428-504 end to end in the mirror, and byte parity of the proof
    stream with the restatement's."""
    g, gp = golden["fri_commit_ref_test"], golden["fri_prove_ref_test"]
    n0, expansion, ncol = g["n0"], g["expansion"], g["ncol"]
    omega = so.primitive_nth_root(n0)
    coeffs = [so.fe_new(i) for i in range(64)]
    codeword = _eval_domain(coeffs, [so.fe_pow(omega, i) for i in range(n0)])
    assert codeword == so.unpack(bytes.fromhex(g["codeword"]))
    # commit alone
    stream = hm.Stream()
    layers = np.zeros((n0 * 2, 2), dtype=np.uint64)
    k = hm.chk(hm.lib.sh_fri_commit(hm.p(hm.arr(codeword)), sz(n0), hm.fe(G), hm.fe(omega), sz(expansion), sz(ncol),
                                    stream.h, hm.p(layers)))
    assert k == len(g["layer_lens"]) and stream.objects == g["objects"]
    # prove
    stream, idx = _prove(codeword, G, omega, expansion, ncol)
    assert idx == gp["indices"]
    assert hashlib.sha256(stream.serialize()).hexdigest() == gp["serialized_sha256"]
    assert [hashlib.sha256(o.encode()).hexdigest() for o in stream.objects] == gp["objects_sha256"]
    ok, points = hm.fri_verify(G, omega, n0, expansion, ncol, hm.Stream.from_objects(stream.objects))
    assert ok
    for i, v in points:
        assert so.poly_eval(coeffs, so.fe_pow(omega, i)) == v
    # the restatement's verifier accepts the mirror's proof
    ps = so.ProofStream.deserialize(stream.serialize())
    assert so.Fri(G, omega, n0, expansion, ncol).verify(ps, [])
    # the reference-shaped commit() + query() give the same stream
    stream2, _ = _prove(codeword, G, omega, expansion, ncol, by_value=True)
    assert stream2.serialize() == stream.serialize()
    # corrupted codeword is rejected
    bad = list(codeword)
    for i in range(63 // 3):
        bad[i] = 0
    stream, _ = _prove(bad, G, omega, expansion, ncol)
    ok, _ = hm.fri_verify(G, omega, n0, expansion, ncol, hm.Stream.from_objects(stream.objects))
    assert ok is False


@pytest.mark.parametrize("log_n,expansion,ncol", [(12, 4, 17), (10, 8, 8), (14, 4, 2)])
def test_fri_prove_byte_parity_coset_codeword(log_n, expansion, ncol):
    n = 1 << log_n
    omega = so.primitive_nth_root(n)
    coeffs = oc.splitmix(n // expansion, 77 + log_n)
    cw = oc.to_ints(oc.coset_lde(coeffs, n, G))
    stream, idx = _prove(cw, G, omega, expansion, ncol)
    ps = so.ProofStream()
    ref_idx = so.Fri(G, omega, n, expansion, ncol).prove(cw, ps)
    assert idx == ref_idx
    assert stream.serialize() == ps.serialize()
    ok, _ = hm.fri_verify(G, omega, n, expansion, ncol, hm.Stream.from_objects(stream.objects))
    assert ok


def test_fri_commit_wrong_order_panics():
    cw = hm.arr(so.splitmix_field_vector(64, seed=1))
    stream = hm.Stream()
    with pytest.raises(hm.HostPanic, match="omega does not have the right order"):
        hm.chk(hm.lib.sh_fri_commit(hm.p(cw), sz(64), hm.fe(G), hm.fe(so.primitive_nth_root(128)), sz(4), sz(2),
                                    stream.h, None))

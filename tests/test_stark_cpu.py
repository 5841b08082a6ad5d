"""Stark::prove's host side without a GPU: the restatement of the polynomial
algebra, multivariate polynomials, Rescue-Prime and stark.rs against the
reference's own known answers and unit-test cases, and the C++ mirror of the
same (stark_amd/host/stark_prover.hpp) against the restatement."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import stark_oracle as so
import stark_prover_oracle as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = so.P


@pytest.fixture(scope="module")
def hm():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stark_amd", "host")])
    import host_mirror
    return host_mirror


@pytest.fixture(scope="module")
def rp():
    return sp.RescuePrime()


# --- rescue_prime.rs ---------------------------------------------------------------------
def test_rescue_parameters_match_reference_tables(rp, ref_kats):
    """The derived parameters equal the literal tables of rescue_prime.rs:29-159."""
    k = ref_kats["rescue_prime"]
    assert rp.round_constants == [int(x) for x in k["round_constants"]]
    assert sum(rp.MDS, []) == [int(x) for x in k["MDS"]]
    assert sum(rp.MDS_inv, []) == [int(x) for x in k["MDS_inv"]]
    assert rp.alpha == int(k["alpha"]) and rp.alpha_inv == int(k["alpha_inv"])
    assert (rp.m, rp.N) == (int(k["m"]), int(k["N"]))


def test_rescue_hash_kats(rp, ref_kats):
    """rescue_prime.rs:415-422."""
    for i, o in ref_kats["rescue_prime"]["hash_kats"]:
        assert rp.hash(int(i)) == int(o)


def test_rescue_trace_and_constraints(rp):
    """rescue_prime.rs:425-484: the trace satisfies boundary and transition constraints and a
    corrupted trace does not."""
    a = 57322816861100832358702415967512842988
    b = 89633745865384635541695204788332415101
    trace = rp.trace(a)
    assert len(trace) == rp.N + 1 and trace[-1][0] == b == rp.hash(a)
    for (cycle, reg, val) in rp.boundary_constraints(b):
        assert trace[cycle][reg] == val
    omicron = so.primitive_nth_root(1 << 119)
    air = rp.transition_constraints(omicron)
    assert [len(p.dict) for p in air] == [272, 272]
    for i in range(rp.N):
        point = [so.fe_pow(omicron, i)] + trace[i] + trace[i + 1]
        assert all(p.eval(point) == 0 for p in air)
    random.seed(1)
    bad = [list(r) for r in trace]
    bad[5][1] = so.fe_add(bad[5][1], 1)
    assert any(p.eval([so.fe_pow(omicron, i)] + bad[i] + bad[i + 1]) != 0 for p in air for i in (4, 5))


# --- multivariate_poly.rs ----------------------------------------------------------------
def test_mpolynomial_evaluate():
    """multivariate_poly.rs:247-268."""
    v = sp.MPolynomial.variables(4)
    c = sp.MPolynomial.constant
    m1 = c(1) * v[0] + c(2) * v[1] + c(5) * v[2].pow(3)
    m2 = c(1) * v[0] * v[3] + c(5) * v[3].pow(3) + c(5)
    m3 = m1 * m2
    point = [0, 5, 5, 2]
    e1, e2, e3 = m1.eval(point), m2.eval(point), m3.eval(point)
    assert so.fe_mul(e1, e2) == e3
    assert so.fe_add(e1, e2) == (m1 + m2).eval(point)
    assert (e1, e2, e3) == (635, 45, 28575)                     # the literal values of :264-266


def test_mpolynomial_lift_and_symbolic(ref_kats):
    """multivariate_poly.rs:271-306 (test_lift): lifting the interpolant to the last of four
    variables and evaluating at (_, _, _, x) gives the polynomial's value."""
    k = ref_kats["lagrange_lift"]
    poly = so.lagrange([int(x) for x in k["x"]], [int(y) for y in k["y"]])
    lifted = sp.MPolynomial.lift(poly, 3)
    for x in (0, 1, 5, 12345, P - 2):
        assert lifted.eval([0, 0, 0, x]) == so.poly_eval(poly, x)
    v = sp.MPolynomial.variables(2)
    m = sp.MPolynomial.constant(3) * v[0].pow(2) * v[1] - v[1].pow(3) + sp.MPolynomial.constant(7)
    pa, pb = [1, 2], [5, 0, 1]                                 # x + 2, 5x^2 + 1
    sym = m.eval_symbolic([pa, pb])
    for x in (0, 3, 99, P - 5):
        assert so.poly_eval(sym, x) == m.eval([so.poly_eval(pa, x), so.poly_eval(pb, x)])


def test_mpolynomial_keeps_cancelled_terms():
    v = sp.MPolynomial.variables(2)
    d = (v[0] * v[1]) - (v[0] * v[1])
    assert d.is_zero() and len(d.dict) == 1                    # the stored term stays (stark.rs:154 iterates it)
    assert d.pow(3).dict == {}


# --- univariate_poly.rs ------------------------------------------------------------------
def test_polynomial_algebra_identities():
    random.seed(3)
    for _ in range(30):
        a = [random.randrange(1, P)] + [random.randrange(P) for _ in range(random.randrange(0, 9))]
        b = [random.randrange(1, P)] + [random.randrange(P) for _ in range(random.randrange(0, 6))]
        x = random.randrange(P)
        ev = so.poly_eval
        assert ev(sp.poly_add(a, b), x) == so.fe_add(ev(a, x), ev(b, x))
        assert ev(sp.poly_sub(a, b), x) == so.fe_sub(ev(a, x), ev(b, x))
        assert ev(sp.poly_mul(a, b), x) == so.fe_mul(ev(a, x), ev(b, x))
        assert sp.poly_div(sp.poly_mul(a, b), b) == a                       # exact division
        assert ev(sp.poly_compose(a, b), x) == ev(a, ev(b, x))
        assert ev(sp.poly_pow(b, 5), x) == so.fe_pow(ev(b, x), 5)
        assert sp.poly_add(sp.poly_neg(a), a) == [0] * len(a)
    z = sp.zeroifier_domain([3, 5, 11])
    assert z == [1, P - 19, 103, P - 165] and all(so.poly_eval(z, d) == 0 for d in (3, 5, 11))
    assert sp.poly_div([1, 0], [1, 0, 0]) == [0]                            # higher-degree divisor (:449-451)
    with pytest.raises(ZeroDivisionError):
        sp.poly_div([1, 2], [0, 0])
    assert sp.poly_pow([0, 0], 3) == [0] and sp.poly_pow([2, 1], 0) == [1]


def test_polynomial_division_reference_case():
    """univariate_poly.rs test_div: (x^2 - 1) / (x - 1) = x + 1, and a quotient with remainder dropped."""
    assert sp.poly_div([1, 0, P - 1], [1, P - 1]) == [1, 1]
    assert sp.poly_div([1, 0, 0, 5], [1, 2]) == [1, P - 2, 4]               # x^3 + 5 = (x + 2)(x^2 - 2x + 4) - 3


# --- stark.rs ----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def rescue_proof(rp):
    inp = so.fe_sample(bytes([48, 120, 100, 101, 97, 100, 98, 101, 101, 102]))    # stark.rs:740
    out = rp.hash(inp)
    st = sp.Stark(4, 2, 2, rp.m, rp.N + 1)
    air = rp.transition_constraints(st.omicron)
    boundary = rp.boundary_constraints(out)
    proof = st.prove(rp.trace(inp), air, boundary, sp.counter_rng(b"stark-test"))
    return st, air, boundary, proof, inp, out


def test_stark_parameters(rescue_proof, ref_kats):
    st = rescue_proof[0]
    assert st.omicron == int(ref_kats["omicron_128th_root"])                # stark.rs:65
    assert (st.num_randomizers, st.omicron_domain_length, st.fri_domain_length) == (8, 128, 512)
    assert st.fri.num_rounds() == 6
    air, boundary = rescue_proof[1], rescue_proof[2]
    assert st.transition_degree_bounds(air) == [105, 105]
    assert st.transition_quotient_degree_bounds(air) == [78, 78]
    assert st.max_degree(air) == 127
    assert st.boundary_quotient_degree_bounds(36, boundary) == [34, 34]


def test_stark_prove_verify_restatement(rescue_proof):
    """stark.rs:731-776 (test_stark): an honest proof verifies."""
    st, air, boundary, proof, _, out = rescue_proof
    assert st.verify(proof, air, boundary)
    ps = so.ProofStream.deserialize(proof)
    # 2 boundary roots + randomizer root + FRI (6 roots + last codeword + 5 layers x (2 + 6)) + openings
    assert len(ps.objects) == 3 + 7 + 5 * 8 + 2 * 8 * 2 + 4 * 2
    wrong = [(c, r, so.fe_add(v, 1) if c else v) for (c, r, v) in boundary]
    assert not st.verify(proof, air, wrong)                                 # a different claimed output


def test_stark_proof_is_a_function_of_the_random_bytes(rescue_proof, rp):
    st, air, boundary, proof, inp, _ = rescue_proof
    again = st.prove(rp.trace(inp), air, boundary, sp.counter_rng(b"stark-test"))
    other = st.prove(rp.trace(inp), air, boundary, sp.counter_rng(b"another seed"))
    assert again == proof and other != proof
    assert st.verify(other, air, boundary)


# --- the C++ mirror (host parts only) ----------------------------------------------------
def test_mirror_rescue_parameters_and_hash(hm, rp, ref_kats):
    rc = np.zeros((108, 2), dtype=np.uint64)
    mds = np.zeros((4, 2), dtype=np.uint64)
    mdi = np.zeros((4, 2), dtype=np.uint64)
    ai = np.zeros(2, dtype=np.uint64)
    hm.chk(hm.lib.sh_rescue_params(hm.p(rc), hm.p(mds), hm.p(mdi), hm.p(ai)))
    k = ref_kats["rescue_prime"]
    assert hm.ints(rc) == [int(x) for x in k["round_constants"]]
    assert hm.ints(mds) == [int(x) for x in k["MDS"]]
    assert hm.ints(mdi) == [int(x) for x in k["MDS_inv"]]
    assert hm.val(ai) == int(k["alpha_inv"])
    out = (C.c_uint64 * 2)()
    for i, o in k["hash_kats"]:
        hm.chk(hm.lib.sh_rescue_hash(hm.fe(int(i)), out))
        assert hm.val(out) == int(o)


def test_mirror_air_matches_restatement(hm, rp):
    omicron = so.primitive_nth_root(128)
    air = rp.transition_constraints(omicron)
    random.seed(11)
    for _ in range(3):
        pt = [random.randrange(P) for _ in range(5)]
        nt = np.zeros(2, dtype=np.uint64)
        vals = np.zeros((2, 2), dtype=np.uint64)
        n = hm.chk(hm.lib.sh_rescue_air(hm.fe(omicron), hm.p(hm.arr(pt)), hm.p(nt), hm.p(vals)))
        assert n == 2 and [int(x) for x in nt] == [len(a.dict) for a in air]
        assert hm.ints(vals) == [a.eval(pt) for a in air]


def _poly_op(hm, op, a, b):
    aa, bb = hm.arr(a), hm.arr(b if b else [0])
    n = C.c_size_t()
    out = np.zeros((len(a) + len(b) + 64 + (len(a) * len(b) if op == 4 else 0) + (len(a) * 8 if op == 5 else 0), 2),
                   dtype=np.uint64)
    hm.chk(hm.lib.sh_poly_op(op, hm.p(aa), C.c_size_t(len(a)), hm.p(bb), C.c_size_t(len(b)), hm.p(out), C.byref(n)))
    return hm.ints(out[:n.value])


def test_mirror_polynomial_algebra(hm):
    random.seed(5)
    for _ in range(25):
        a = [random.randrange(P) for _ in range(random.randrange(1, 10))]
        b = [random.randrange(1, P)] + [random.randrange(P) for _ in range(random.randrange(0, 6))]
        assert _poly_op(hm, 0, a, b) == sp.poly_add(a, b)
        assert _poly_op(hm, 1, a, b) == sp.poly_sub(a, b)
        assert _poly_op(hm, 2, a, b) == sp.poly_mul(a, b)
        assert _poly_op(hm, 3, a, b) == sp.poly_div(a, b)                   # includes leading-zero dividends
        assert _poly_op(hm, 4, a, b) == sp.poly_compose(a, b)
        assert _poly_op(hm, 5, a, [5, 0]) == sp.poly_pow(a, 5)
        assert _poly_op(hm, 6, a, []) == sp.zeroifier_domain(a)
        assert _poly_op(hm, 7, a, []) == sp.poly_neg(a)
    assert _poly_op(hm, 3, [0, 0, 7, 1, 2], [1, 3]) == sp.poly_div([0, 0, 7, 1, 2], [1, 3])
    with pytest.raises(hm.HostPanic, match="divide by zero polynomial"):
        _poly_op(hm, 3, [1, 2], [0, 0])


# --- on-disk formats (univariate_poly.rs:224-238, proof_stream.rs:36-47) -----------------
def test_polynomial_file_format(hm, tmp_path):
    coeffs = [0, 1, P - 1, 1 << 64, (1 << 96) + 5, 12345]
    text = sp.polynomial_to_json(coeffs)
    assert text.startswith('{"coeffs":[{"value":[0,[]]},{"value":[1,[1]]},')
    assert sp.polynomial_from_json(text) == coeffs
    path = str(tmp_path / "poly.json").encode()
    out = np.zeros((len(coeffs), 2), dtype=np.uint64)
    n = C.c_size_t()
    hm.chk(hm.lib.sh_poly_save_load(hm.p(hm.arr(coeffs)), C.c_size_t(len(coeffs)), path, hm.p(out), C.byref(n)))
    assert hm.ints(out[:n.value]) == coeffs
    assert open(path, "rb").read().decode() == text             # the mirror writes the same bytes


def test_proof_file_roundtrip(hm, rescue_proof, tmp_path):
    """A proof is ProofStream::serialize's bytes: written to disk and read back it deserialises to
    the same objects in both implementations."""
    proof = rescue_proof[3]
    f = tmp_path / "proof.bin"
    f.write_bytes(proof)
    back = f.read_bytes()
    objs = so.ProofStream.deserialize(back).objects
    assert hm.Stream.deserialize(back).objects == objs
    assert hm.Stream.from_objects(objs).serialize() == proof


def test_reference_polynomial_known_answers(hm):
    """The literal cases of the reference's own unit tests (univariate_poly.rs:508-817), on the
    restatement and on the C++ mirror."""
    ev = so.poly_eval
    cases = [
        # (op, a, b, expected)                                                     reference test
        (0, [10, 3, 1], [90, 3, 1], [100, 6, 2]),                                   # test_add_same_degree
        (1, [80, 6, 1], [40, 3, 6], [40, 3, P - 5]),                                # test_sub_same_degree
        (1, [80, 6, 1, 80, 6, 1], [40, 3, 6], [80, 6, 1, 40, 3, P - 5]),            # test_sub_larger_lhs
        (1, [40, 3, 6], [80, 6, 1, 80, 6, 1], [P - 80, P - 6, P - 1, P - 40, P - 3, 5]),   # test_sub_larger_rhs
        (0, [10, 3, 1], [1, 2, 3, 90, 3, 1], [1, 2, 3, 100, 6, 2]),                 # test_add_diff_degree
        (3, [1, 3, 2], [1, 2], [1, 1]),                                             # test_div_1: (x+2)(x+1) / (x+2)
        (3, [1, 3, 2], [1, 1], [1, 2]),
    ]
    fn = {0: sp.poly_add, 1: sp.poly_sub, 2: sp.poly_mul, 3: sp.poly_div}
    for op, a, b, want in cases:
        assert fn[op](a, b) == want
        assert _poly_op(hm, op, a, b) == want
    prod = sp.poly_mul([10, 1, 0, 1], [3, 1, 17])                                   # test_mul
    assert ev(prod, 2) == 2635 and _poly_op(hm, 2, [10, 1, 0, 1], [3, 1, 17]) == prod
    assert ev(sp.poly_neg([10, 3, 1]), 9) == P - 838                                # test_negate
    assert _poly_op(hm, 7, [10, 3, 1], []) == sp.poly_neg([10, 3, 1])
    assert ev([10, 3, 1], 2) == 47                                                  # test_eval
    assert ev([1, 0], 2) == 2 and ev([3, 5], 2) == 11 and ev([1, 2, 3], 2) == 11    # test_polynomial_coefficient_order
    # test_distributivity
    a, b, c = [0, 1, 2], [2, 2, 1], [0, 5, 2, 5, 5, 1]
    assert sp.poly_mul(a, sp.poly_add(b, c)) == sp.poly_add(sp.poly_mul(a, b), sp.poly_mul(a, c))
    # test_div_2: (x^6 + 20x^5 + 35x^4 - 1460x^3 - 9396x^2 - 12960x) / (x^2 - 7x - 18), with the
    # reference's negative literals mapped into the field
    num = [so.fe_new(v) for v in (1, 20, 35, -1460, -9396, -12960, 0)]
    den = [so.fe_new(v) for v in (1, -7, -18)]
    q = sp.poly_div(num, den)
    assert q == _poly_op(hm, 3, num, den) and len(q) == 5
    assert sp.poly_mul(q, den) == num                                               # the division is exact
    # test_polynomial_division_consistency: quotient * divisor + remainder = dividend
    q = sp.poly_div([3, 2, 1], [1, 1])
    rem = sp.poly_sub([3, 2, 1], sp.poly_mul(q, [1, 1]))
    assert sp.poly_add(sp.poly_mul(q, [1, 1]), rem) == [3, 2, 1] and q == [3, P - 1]
    assert so.test_colinearity([(0, 0), (1, 1), (2, 2)])                            # test_test_colinearity


# --- proof_stream.rs: the transcript's hash ---------------------------------------------
def test_mirror_shake256_vs_hashlib(hm):
    """SHAKE256 of the mirror (proof_stream.rs:52,63: sha3::Shake256) against hashlib: message lengths on both sides
    of the 136-byte rate, several blocks, and outputs longer than one block."""
    import hashlib
    lib = hm.lib
    rnd = random.Random(202)
    for n in (0, 1, 7, 8, 135, 136, 137, 271, 272, 273, 1000, 1360, 4097):
        data = bytes(rnd.randrange(256) for _ in range(n))
        for outlen in (1, 32, 135, 136, 137, 300):
            out = (C.c_uint8 * outlen)()
            lib.sh_shake256(data, C.c_size_t(n), out, C.c_size_t(outlen))
            assert bytes(out) == hashlib.shake_256(data).digest(outlen), (n, outlen)


def test_mirror_fiat_shamir_is_shake256_of_the_serialised_stream(hm):
    """prover_fiat_shamir (proof_stream.rs:50-58) over a stream that grows by one object a round, as in Fri::commit."""
    import hashlib
    ps = hm.Stream()
    objs = []
    for r in range(20):
        objs.append("%064x" % (r * 0x9E3779B97F4A7C15))
        ps.push(objs[-1])
        ser = len(objs).to_bytes(8, "little") + b"".join(len(o).to_bytes(8, "little") + o.encode() for o in objs)
        assert bytes(ps.serialize()) == ser
        assert bytes(ps.fiat_shamir(32)) == hashlib.shake_256(ser).digest(32)

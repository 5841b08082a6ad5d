"""The long-trace prover on the device (stark_amd/host/stark_scaled.hpp above the C ABI) and the
device passes it is built from, against the CPU restatements: every new entry point element for
element, proofs byte for byte at 28 .. 2^16 rows, and at 2^20 rows both verifiers."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_c as oc
import stark_oracle as so
import stark_prover_oracle as sp
import stark_scaled_oracle as ss

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, G = so.P, so.GENERATOR


@pytest.fixture(scope="module")
def ctx():
    from stark_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def hm():
    import host_mirror
    hm_ = host_mirror
    hm_.lib.sh_chain_new.restype = C.c_void_p
    hm_.lib.sh_chain_random_bytes.restype = C.c_size_t
    hm_.lib.sh_chain_fri_domain_length.restype = C.c_size_t
    return hm_


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def _vec(n, seed, zeros=()):
    x = oc.splitmix(n, seed)
    for i in zeros:
        x[i] = 0
    return x


# ---- the device passes ----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 256, 2049, 4096, 4097, 1 << 16])
def test_vec_inv(ctx, n):
    """FieldElement::inverse (field.rs:67-93) pointwise, zero to zero."""
    x = _vec(n, 40 + n, zeros=[0, n // 2] if n > 2 else [])
    x[n - 1] = oc.fe(P - 1)
    d = _dev(x)
    out = _dev(np.zeros_like(x))
    ctx.vec_inv_dev(d, out, n)
    ctx.synchronize()
    got = oc.to_ints(_host(out))
    want = [so.fe_inverse(v) if v else 0 for v in oc.to_ints(x)]
    assert got == want
    ctx.vec_inv_dev(d, d, n)                             # in place
    ctx.synchronize()
    assert oc.to_ints(_host(d)) == want


@pytest.mark.parametrize("n,cols,pad", [(1, 1, 0), (300, 3, 5), (1 << 12, 2, 0), (70000, 1, 0)])
def test_poly_compose_linear(ctx, n, cols, pad):
    """Polynomial::compose with s*x (univariate_poly.rs:203-221): coefficient j times s^j."""
    s = so.fe_pow(G, 12345)
    stride = n + pad
    x = _vec(cols * stride, 50 + n)
    d = _dev(x)
    ctx.poly_compose_linear_dev(d, n, s, cols, stride)
    ctx.synchronize()
    got = oc.to_ints(_host(d))
    src = oc.to_ints(x)
    for c in range(cols):
        w = 1
        for j in range(n):
            assert got[c * stride + j] == src[c * stride + j] * w % P
            w = w * s % P
        assert got[c * stride + n:(c + 1) * stride] == src[c * stride + n:(c + 1) * stride]     # padding untouched


@pytest.mark.parametrize("log_n,cols", [(0, 1), (3, 2), (10, 3), (13, 2), (18, 1)])
def test_coset_interpolate_inverts_coset_lde(ctx, log_n, cols):
    """Polynomial::lagrange over offset * omega^i (univariate_poly.rs:127-177): the inverse of the coset transform,
    and equal to the restatement's interpolation."""
    import torch
    n = 1 << log_n
    offset = so.fe_pow(G, 3)
    coeffs = _vec(cols * n, 60 + log_n).reshape(cols, n, 2)
    d = _dev(coeffs)
    vals = torch.empty_like(d)
    torch.cuda.synchronize()
    ctx.coset_lde_batch_dev(d, n, n, cols, n, offset, vals, n)
    ctx.synchronize()
    for c in range(cols):
        assert np.array_equal(_host(vals[c]), oc.coset_lde(coeffs[c], n, offset))
    ctx.coset_interpolate_batch_dev(vals, n, offset, cols, n)
    ctx.synchronize()
    assert np.array_equal(_host(vals), coeffs)
    if log_n <= 10:
        v = oc.to_ints(oc.coset_lde(coeffs[0], n, offset))
        assert ss.coset_interpolate(v, offset) == oc.to_ints(coeffs[0])


def test_poly_lincomb(ctx):
    """Weighted sum of polynomials times powers of x (stark.rs:385-404)."""
    n_out = 5000
    lens, shifts = [5000, 1200, 1200, 1, 0], [0, 0, 3800, 4999, 17]
    weights = [so.fe_pow(G, 7 + k) for k in range(5)]
    weights[1] = P - 1
    srcs = [_vec(max(l, 1), 70 + k) for k, l in enumerate(lens)]
    devs = [_dev(s) for s in srcs]
    out = _dev(np.zeros((n_out, 2), dtype=np.uint64))
    ctx.poly_lincomb_dev([(devs[k], lens[k], shifts[k], weights[k]) for k in range(5)], out, n_out)
    ctx.synchronize()
    want = [0] * n_out
    for k in range(5):
        for j, c in enumerate(oc.to_ints(srcs[k])[:lens[k]]):
            want[j + shifts[k]] = (want[j + shifts[k]] + weights[k] * c) % P
    assert oc.to_ints(_host(out)) == want
    # accumulate in place: out += w * src
    ctx.poly_lincomb_dev([(out, n_out, 0, 1), (devs[0], 100, 50, 5)], out, n_out)
    ctx.synchronize()
    for j, c in enumerate(oc.to_ints(srcs[0])[:100]):
        want[j + 50] = (want[j + 50] + 5 * c) % P
    assert oc.to_ints(_host(out)) == want
    from stark_amd import StarkError
    with pytest.raises(StarkError, match="fit the output"):
        ctx.poly_lincomb_dev([(devs[0], 5000, 1, 1)], out, n_out)
    with pytest.raises(StarkError, match="16 terms"):
        ctx.poly_lincomb_dev([(devs[0], 10, 0, 1)] * 17, out, n_out)


def test_mpoly_eval_rescue_constraints(ctx):
    """MPolynomial::eval (multivariate_poly.rs:42-72) of the Rescue-Prime constraints at n points, with the
    rotation that stands for the composition with omicron * x and the pointwise factor."""
    rp = sp.RescuePrime()
    air = ss.rescue_column_air(rp)
    n, nv = 1000, 9
    cols = [_vec(n, 80 + v) for v in range(7)]            # x, cur0, cur1, k0..k3 (next = cur rotated)
    mul = _vec(n, 99)
    d = [_dev(c) for c in cols]
    dm = _dev(mul)
    ptrs = [d[0], d[1], d[2], d[1], d[2], d[3], d[4], d[5], d[6]]
    rot = [0, 0, 0, 3, 3, 0, 0, 0, 0]
    ints = [oc.to_ints(c) for c in cols]
    mi = oc.to_ints(mul)
    for a in air:
        terms = [(c, k) for k, c in a.dict.items() if c]
        assert len(terms) > 20 and all(len(k) <= nv for _, k in terms)
        out = _dev(np.zeros((n, 2), dtype=np.uint64))
        ctx.mpoly_eval_dev(ptrs, rot, terms, out, n)
        ctx.synchronize()                                 # the call only queues (its table goes up through a staging slot)
        got = oc.to_ints(_host(out))
        ctx.mpoly_eval_dev(ptrs, rot, terms, out, n, mul_ptr=dm)
        ctx.synchronize()
        got_mul = oc.to_ints(_host(out))
        for i in list(range(0, n, 37)) + [n - 3, n - 2, n - 1]:
            point = [ints[0][i], ints[1][i], ints[2][i], ints[1][(i + 3) % n], ints[2][(i + 3) % n]] + \
                    [ints[3 + k][i] for k in range(4)]
            want = a.eval(point)
            assert got[i] == want and got_mul[i] == want * mi[i] % P
    # exponents beyond the AIR's: x^200 * y^2 + 5
    out = _dev(np.zeros((n, 2), dtype=np.uint64))
    ctx.mpoly_eval_dev([d[0], d[1]], None, [(1, (200, 2)), (5, (0, 0))], out, n)
    ctx.synchronize()
    got = oc.to_ints(_host(out))
    for i in (0, 1, n - 1):
        assert got[i] == (pow(ints[0][i], 200, P) * pow(ints[1][i], 2, P) + 5) % P


def test_poly_degree(ctx):
    """Polynomial::degree (univariate_poly.rs:69-85) on lowest-first coefficients."""
    import torch
    n = 1 << 16
    x = np.zeros((n, 2), dtype=np.uint64)
    d = _dev(x)
    assert ctx.poly_degree_dev(d, n) == (0, True)
    # the posted form: 0 for the zero polynomial, else degree + 1, left on the device
    word = torch.full((3,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.poly_degree_post_dev(d, n, word[0:])
    y = x.copy()
    y[12345] = oc.fe(7)
    d2 = _dev(y)
    ctx.poly_degree_post_dev(d2, n, word[1:])
    y[0] = oc.fe(1)
    ctx.poly_degree_post_dev(_dev(y[:1]), 1, word[2:])
    ctx.synchronize()
    assert word.cpu().tolist() == [0, 12346, 1]
    x[0] = oc.fe(9)
    assert ctx.poly_degree_dev(_dev(x), n) == (0, False)
    x[40000, 1] = 1                                       # a value whose low word is zero
    x[123] = oc.fe(5)
    assert ctx.poly_degree_dev(_dev(x), n) == (40000, False)
    x[n - 1] = oc.fe(1)
    assert ctx.poly_degree_dev(_dev(x), n) == (n - 1, False)
    assert ctx.poly_degree_dev(_dev(x), 40001) == (40000, False)


@pytest.mark.parametrize("n,polys,pad", [(1, 1, 0), (15, 2, 3), (16, 1, 0), (4096, 3, 0), (4097, 2, 1), (100000, 2, 7),
                                         (1 << 20, 1, 0)])
def test_poly_eval_points(ctx, n, polys, pad):
    """Polynomial::eval (univariate_poly.rs:87-96) of resident polynomials at a few points, against Horner
    on Python integers."""
    from stark_amd import StarkError
    stride = n + pad
    c = oc.splitmix(stride * polys, 60 + n % 97).reshape(polys, stride, 2)
    points = [0, 1, P - 1, G, 123456789, pow(G, 77, P)]
    got = ctx.poly_eval_points_dev(_dev(c), n, points, n_polys=polys, stride=stride)
    for q in range(polys):
        coeffs = oc.to_ints(np.ascontiguousarray(c[q, :n]))
        for k, x in enumerate(points):
            acc = 0
            for v in reversed(coeffs):
                acc = (acc * x + v) % P
            assert got[q][k] == acc, (q, k)
    with pytest.raises(StarkError, match="at most 16 points"):
        ctx.poly_eval_points_dev(_dev(c), n, list(range(17)), n_polys=polys, stride=stride)
    with pytest.raises(StarkError, match="not reduced"):
        ctx.poly_eval_points_dev(_dev(c), n, [P], n_polys=polys, stride=stride)


@pytest.mark.parametrize("width,n", [(17, 1000), (32, 257), (1, 5), (40, 64), (14, 70), (15, 300), (16, 300), (30, 65), (45, 33)])
def test_sample_bytes(ctx, width, n):
    """FieldElement::sample (field.rs:110-116) of n byte strings; reversed as Polynomial::coeffs is filled."""
    import torch
    raw = hashlib.shake_256(b"sample%d" % width).digest(width * n)
    db = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    out = _dev(np.zeros((n, 2), dtype=np.uint64))
    want = [so.fe_sample(raw[i * width:(i + 1) * width]) for i in range(n)]
    ctx.sample_bytes_dev(db, width, n, out)
    ctx.synchronize()
    assert oc.to_ints(_host(out)) == want
    ctx.sample_bytes_dev(db, width, n, out, reversed_=True)
    ctx.synchronize()
    assert oc.to_ints(_host(out)) == want[::-1]


@pytest.mark.parametrize("n_table,m,cols", [(36, 36, 2), (36, 27, 4), (1, 1, 1), (1000, 999, 1), (5000, 4104, 2)])
def test_geometric_interpolation(hm, n_table, m, cols):
    """Polynomial::lagrange over omicron^0 .. omicron^(m-1) and the zeroifier of those points
    (univariate_poly.rs:127-164,254-264) through the device-backed GeometricDomain."""
    q = so.primitive_nth_root(1 << (2 * n_table).bit_length())
    ys = so.splitmix_field_vector(m * cols, seed=n_table + m)
    out = np.zeros((m * cols, 2), dtype=np.uint64)
    z = np.zeros((m + 1, 2), dtype=np.uint64)
    hm.chk(hm.lib.sh_geometric_interpolate(hm.fe(q), C.c_size_t(n_table), hm.p(hm.arr(ys)), C.c_size_t(m),
                                           C.c_size_t(cols), hm.p(out), hm.p(z)))
    geo = ss.GeometricDomain(q, n_table)
    got = hm.ints(out)
    for c in range(cols):
        assert got[c * m:(c + 1) * m] == geo.interpolate(ys[c * m:(c + 1) * m])
    assert hm.ints(z) == geo.zeroifier(m)
    if m <= 36:
        xs = [so.fe_pow(q, i) for i in range(m)]
        want = so.lagrange_traditional(xs, ys[:m])[::-1]
        assert got[:m] == (want + [0] * m)[:m]


# ---- whole proofs ---------------------------------------------------------------------------------
class _Chain:
    def __init__(self, hm, inp, rows, expansion=4, ncol=2, security=2):
        self.hm = hm
        self.h = C.c_void_p(hm.lib.sh_chain_new(hm.fe(inp), C.c_size_t(rows), C.c_size_t(expansion), C.c_size_t(ncol),
                                                C.c_size_t(security)))
        if not self.h:
            raise hm.HostPanic(hm.lib.sh_last_panic().decode())
        out = (C.c_uint64 * 2)()
        hm.chk(hm.lib.sh_chain_output(self.h, out))
        self.output = hm.val(out)
        self.random_bytes = hm.lib.sh_chain_random_bytes(self.h)

    def prove(self, random: bytes):
        hm = self.hm
        rb = (C.c_uint8 * len(random)).from_buffer_copy(random)
        proof = (C.c_uint8 * (1 << 20))()
        plen = C.c_size_t(len(proof))
        tm = (C.c_double * 8)()
        used = hm.chk(hm.lib.sh_chain_prove(self.h, rb, C.c_size_t(len(random)), proof, C.byref(plen), tm))
        self.timings = dict(zip(("host", "interpolate", "quotients", "lde", "merkle", "fri", "openings", "total"), tm))
        return bytes(proof[:plen.value]), used

    def verify(self, output: int, proof: bytes) -> int:
        pb = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        return self.hm.lib.sh_chain_verify(self.h, self.hm.fe(output), pb, C.c_size_t(len(proof)))

    def free(self):
        if self.h:
            self.hm.lib.sh_chain_free(self.h)
            self.h = None


@pytest.mark.parametrize("seed,inp", [(b"stark-test", 12345), (b"second", 1)])
def test_proof_equals_reference_algorithm_at_28_rows(hm, seed, inp):
    """The reference's own configuration (stark.rs:731-777): the device prover's bytes are those of the
    line-by-line restatement of the reference's algorithm, and every verifier accepts them."""
    rp = sp.RescuePrime()
    st = sp.Stark(4, 2, 2, rp.m, rp.N + 1)
    air = rp.transition_constraints(st.omicron)
    out = rp.hash(inp)
    random = hashlib.shake_256(seed).digest(17 * (16 + 128))
    want = st.prove(rp.trace(inp), air, rp.boundary_constraints(out), ss.stream_rng(seed, len(random)))
    job = _Chain(hm, inp, rp.N + 1)
    try:
        assert job.output == out and job.random_bytes == len(random)
        got, used = job.prove(random)
        assert used == len(random)
        assert got == want
        assert job.verify(out, got) == 1
        assert st.verify(got, air, rp.boundary_constraints(out))
    finally:
        job.free()


def _golden(log_rows):
    with open(os.path.join(ROOT, "tests", "golden", "scaled_stark.json")) as f:
        return json.load(f).get(str(log_rows))


@pytest.mark.parametrize("log_rows", [5, 6, 8, 10, 12, 14, 16, 18, 20])
def test_proof_bytes_equal_restatement(hm, log_rows):
    """Proofs of the chained workload against the restatement's committed vectors (length and SHA-256
    of the proof, tests/golden/scaled_stark.json); up to 2^12 rows the restatement also runs here and
    the bytes are compared directly, and the restatement's verifier checks the device's proof."""
    g = _golden(log_rows)
    if g is None:
        pytest.skip("no committed vector for this size")
    rows = 1 << log_rows
    job = _Chain(hm, int(g["input"]), rows, g["expansion"], g["ncol"], g["security"])
    try:
        assert job.output == int(g["output"]) and job.random_bytes == g["random_bytes"]
        random = hashlib.shake_256(g["seed"].encode()).digest(g["random_bytes"])
        proof, used = job.prove(random)
        assert used == len(random)
        assert len(proof) == g["proof_len"]
        assert hashlib.sha256(proof).hexdigest() == g["proof_sha256"]
        assert job.verify(job.output, proof) == 1
        assert job.verify((job.output + 1) % P, proof) == 0
        if log_rows <= 12:
            rp = sp.RescuePrime()
            st = ss.ScaledStark(g["expansion"], g["ncol"], g["security"], rp.m, rows)
            trace = ss.rescue_chain_trace(rp, int(g["input"]), rows)
            air, cols = ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows)
            boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
            assert proof == st.prove(trace, air, cols, boundary, ss.stream_rng(g["seed"].encode(), len(random)))
            assert st.verify(proof, air, cols, boundary)
    finally:
        job.free()


VARIANTS = ["rows=2,e=4,c=2,in=5", "rows=3,e=4,c=2,in=5", "rows=100,e=8,c=3,in=7", "rows=513,e=8,c=4,in=11",
            "rows=1000,e=4,c=4,in=99", "rows=1000,e=4,c=5,in=99", "rows=3000,e=16,c=2,in=3", "rows=65537,e=4,c=2,in=8"]


@pytest.mark.parametrize("key", VARIANTS)
def test_proof_bytes_other_lengths_and_parameters(hm, key):
    """Trace lengths that are not powers of two (down to the minimum of two rows), other expansion factors,
    numbers of colinearity tests and inputs, against the restatement's committed vectors.  One case
    (1000 rows, 5 colinearity tests) sits where stark.rs:44-58, which sizes the domains for constraints of
    degree 2, leaves the FRI domain too small for the degree-3 Rescue-Prime constraints: the restatement's
    verifier rejects the restatement's own proof there, and the device prover and the mirror's verifier must do
    the same with the same bytes."""
    g = _golden(key)
    rows = g["rows"]
    job = _Chain(hm, int(g["input"]), rows, g["expansion"], g["ncol"], g["security"])
    try:
        assert job.output == int(g["output"]) and job.random_bytes == g["random_bytes"]
        assert hm.lib.sh_chain_fri_domain_length(job.h) == g["fri_domain"]
        random = hashlib.shake_256(g["seed"].encode()).digest(g["random_bytes"])
        proof, used = job.prove(random)
        assert used == len(random)
        assert len(proof) == g["proof_len"]
        assert hashlib.sha256(proof).hexdigest() == g["proof_sha256"]
        assert job.verify(job.output, proof) == (1 if g["verifies"] else 0)
        assert job.verify((job.output + 1) % P, proof) == 0
    finally:
        job.free()


def test_2_20_rows_both_verifiers(hm):
    """The scaled workload: a 2^20-row Rescue-Prime chain, FRI domain 2^24.  The mirror's verifier and the
    Python restatement's verifier (which evaluates the constant columns through the Lagrange basis
    instead of coefficients) accept the device's proof; a proof for another output is rejected."""
    rp = sp.RescuePrime()
    rows = 1 << 20
    job = _Chain(hm, 2024, rows)
    try:
        assert hm.lib.sh_chain_fri_domain_length(job.h) == 1 << 24
        random = hashlib.shake_256(b"2^20 rows").digest(job.random_bytes)
        proof, used = job.prove(random)
        assert used == len(random)
        print("prove 2^20 rows:", {k: round(v, 1) for k, v in job.timings.items()}, "ms; proof", len(proof), "bytes")
        assert job.verify(job.output, proof) == 1
        assert job.verify((job.output + 5) % P, proof) == 0
        st = ss.ScaledStark(4, 2, 2, rp.m, rows)
        air, cols = ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows)
        assert st.verify(proof, air, cols, ss.rescue_chain_boundary(rp, rows, job.output))
    finally:
        job.free()


def test_more_than_sixteen_opened_points(hm):
    """Nine colinearity tests open eighteen points: the verifier's evaluation of the constant columns' polynomials
    takes them in runs of sixteen."""
    job = _Chain(hm, 5, 64, 4, 9, 2)
    try:
        random = hashlib.shake_256(b"nine tests").digest(job.random_bytes)
        proof, _ = job.prove(random)
        assert job.verify(job.output, proof) == 1
        assert job.verify((job.output + 1) % P, proof) == 0
        rp = sp.RescuePrime()
        st = ss.ScaledStark(4, 9, 2, rp.m, 64)
        assert st.verify(proof, ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, 64),
                         ss.rescue_chain_boundary(rp, 64, job.output))
    finally:
        job.free()


def test_prover_refuses_what_it_cannot_prove(hm):
    """stark.rs:379-380 asserts the degrees of the transition quotients: with one trace value off by one
    the constraints do not vanish on the trace and the quotient is not a polynomial of the bound's degree.
    The prover also stops when its random bytes run out."""
    job = _Chain(hm, 3, 64)
    try:
        rb = (C.c_uint8 * 16)()
        plen = C.c_size_t(0)
        rc = hm.lib.sh_chain_prove(job.h, rb, C.c_size_t(16), None, C.byref(plen), None)
        assert rc < 0 and b"random bytes exhausted" in hm.lib.sh_last_panic()
        random = hashlib.shake_256(b"x").digest(job.random_bytes)
        proof, _ = job.prove(random)
        assert job.verify(job.output, proof) == 1
        hm.lib.sh_chain_corrupt(job.h, C.c_size_t(30), C.c_size_t(1))
        with pytest.raises(hm.HostPanic, match="tq_degrees"):
            job.prove(random)
    finally:
        job.free()


def test_repeated_proofs_do_not_grow_device_memory(hm):
    """Every device vector of a proof goes back to the context's block cache: after the first proofs, more
    proofs and verifications of the same size leave the device's free memory where it was."""
    import torch
    job = _Chain(hm, 77, 1 << 10)
    try:
        random = hashlib.shake_256(b"again").digest(job.random_bytes)
        for _ in range(3):
            proof, _ = job.prove(random)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(40):
            again, _ = job.prove(random)
            assert again == proof
            assert job.verify(job.output, again) == 1
        torch.cuda.synchronize()
        assert free_before - torch.cuda.mem_get_info()[0] < (64 << 20)
    finally:
        job.free()


def test_prepared_and_unprepared_constant_polynomials_agree(hm):
    """The constant columns' polynomials made once with the statement, or interpolated inside every prove
    and verify call: the same proof bytes and the same verdicts."""
    job = _Chain(hm, 4242, 1 << 11)
    try:
        random = hashlib.shake_256(b"prepared").digest(job.random_bytes)
        proof, _ = job.prove(random)
        assert job.verify(job.output, proof) == 1
        hm.lib.sh_chain_use_prepared_constants(job.h, 0)
        again, _ = job.prove(random)
        assert again == proof
        assert job.verify(job.output, proof) == 1
        assert job.verify((job.output + 1) % P, proof) == 0
    finally:
        job.free()

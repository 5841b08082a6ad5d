"""The restatement of the long-trace prover (oracle/stark_scaled_oracle.py) against the line-by-line
restatement of the reference's own algorithm (oracle/stark_prover_oracle.py) and against the
committed vectors; CPU only."""
import hashlib
import json
import os

import pytest

import stark_oracle as so
import stark_prover_oracle as sp
import stark_scaled_oracle as ss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = so.P


@pytest.fixture(scope="module")
def rp():
    return sp.RescuePrime()


def test_geometric_interpolation_equals_lagrange_traditional():
    """univariate_poly.rs:147-164 on the first m powers of a root of unity, m not a power of two."""
    q = so.primitive_nth_root(64)
    geo = ss.GeometricDomain(q, 20)
    for m in (1, 2, 5, 13, 20):
        ys = so.splitmix_field_vector(m, seed=m)
        xs = [so.fe_pow(q, i) for i in range(m)]
        want = so.lagrange_traditional(xs, ys)[::-1]                  # highest first -> lowest first
        want = want + [0] * (m - len(want))
        assert geo.interpolate(ys) == want[:m]
        assert geo.zeroifier(m) == sp.zeroifier_domain(xs)[::-1]      # univariate_poly.rs:254-264
        x = 123456789
        basis = geo.lagrange_basis_at(m, x)
        assert sum(b * y for b, y in zip(basis, ys)) % P == ss.poly_eval(want, x)


def test_divide_exact_and_products():
    a = so.splitmix_field_vector(40, seed=1)
    b = so.splitmix_field_vector(23, seed=2)
    prod = ss.poly_mul(a, b)
    assert prod == so._poly_mul(a, b)                                 # schoolbook, univariate_poly.rs:403-420
    assert ss.divide_exact(prod, b, so.GENERATOR) == a
    with pytest.raises(AssertionError):
        ss.divide_exact(prod[:-1] + [(prod[-1] + 1) % P], b, so.GENERATOR)


def test_chain_reduces_to_the_reference_workload(rp):
    t = ss.rescue_chain_trace(rp, 777, rp.N + 1)
    assert t == rp.trace(777) and t[-1][0] == rp.hash(777)
    assert ss.rescue_chain_boundary(rp, rp.N + 1, t[-1][0]) == rp.boundary_constraints(t[-1][0])
    # substituting the column polynomials for the constant variables gives the reference's constraints
    st = sp.Stark(4, 2, 2, rp.m, rp.N + 1)
    ref_air = rp.transition_constraints(st.omicron)
    cols = ss.rescue_constant_columns(rp, rp.N + 1)
    geo = ss.GeometricDomain(st.omicron, rp.N)
    polys = [geo.interpolate(c) for c in cols]
    x = 987654321
    point = [x, 11, 22, 33, 44]
    col_air = ss.rescue_column_air(rp)
    for a_ref, a_col in zip(ref_air, col_air):
        assert a_ref.eval(point) == a_col.eval(point + [ss.poly_eval(c, x) for c in polys])


@pytest.mark.parametrize("seed,inp", [(b"stark-test", 12345), (b"second", 1)])
def test_proof_bytes_equal_reference_algorithm_at_28_rows(rp, seed, inp):
    """At the reference's own size the fast algebra gives the proof of the reference's algorithm."""
    rows = rp.N + 1
    st = sp.Stark(4, 2, 2, rp.m, rows)
    air = rp.transition_constraints(st.omicron)
    out = rp.hash(inp)
    want = st.prove(rp.trace(inp), air, rp.boundary_constraints(out), ss.stream_rng(seed, 17 * (16 + 128)))
    s2 = ss.ScaledStark(4, 2, 2, rp.m, rows)
    assert (s2.omicron, s2.omega, s2.fri_domain_length) == (st.omicron, st.omega, st.fri_domain_length)
    cair = ss.rescue_column_air(rp)
    assert s2.transition_degree_bounds(cair) == st.transition_degree_bounds(air)
    assert s2.max_degree(cair) == st.max_degree(air) == 127
    cols = ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, out)
    got = s2.prove(rp.trace(inp), cair, cols, boundary, ss.stream_rng(seed, 17 * (16 + 128)))
    assert got == want
    assert s2.verify(got, cair, cols, boundary)
    assert st.verify(got, air, rp.boundary_constraints(out))          # the reference-shaped verifier accepts it too


@pytest.mark.parametrize("log_rows", [5, 6, 8, 10])
def test_committed_vectors(rp, log_rows):
    """tests/golden/scaled_stark.json (written by tests/golden/gen_scaled_stark.py) is reproduced."""
    with open(os.path.join(ROOT, "tests", "golden", "scaled_stark.json")) as f:
        g = json.load(f)[str(log_rows)]
    rows = 1 << log_rows
    st = ss.ScaledStark(g["expansion"], g["ncol"], g["security"], rp.m, rows)
    trace = ss.rescue_chain_trace(rp, int(g["input"]), rows)
    assert trace[-1][0] == int(g["output"])
    air, cols = ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    assert st.fri_domain_length == g["fri_domain"] and st.max_degree(air) == g["max_degree"]
    proof = st.prove(trace, air, cols, boundary, ss.stream_rng(g["seed"].encode(), g["random_bytes"]))
    assert len(proof) == g["proof_len"] and hashlib.sha256(proof).hexdigest() == g["proof_sha256"]


@pytest.mark.parametrize("key", ["rows=2,e=4,c=2,in=5", "rows=3,e=4,c=2,in=5", "rows=100,e=8,c=3,in=7",
                                 "rows=513,e=8,c=4,in=11", "rows=1000,e=4,c=5,in=99"])
def test_committed_vectors_other_lengths_and_parameters(rp, key):
    """The vectors for trace lengths that are not powers of two and for other parameters are reproduced, the
    one the verifier must reject included (FRI domain too small for the constraints' degree, stark.rs:44-58)."""
    with open(os.path.join(ROOT, "tests", "golden", "scaled_stark.json")) as f:
        g = json.load(f)[key]
    rows = g["rows"]
    st = ss.ScaledStark(g["expansion"], g["ncol"], g["security"], rp.m, rows)
    trace = ss.rescue_chain_trace(rp, int(g["input"]), rows)
    assert trace[-1][0] == int(g["output"])
    air, cols = ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    assert st.fri_domain_length == g["fri_domain"] and st.max_degree(air) == g["max_degree"]
    proof = st.prove(trace, air, cols, boundary, ss.stream_rng(g["seed"].encode(), g["random_bytes"]))
    assert len(proof) == g["proof_len"] and hashlib.sha256(proof).hexdigest() == g["proof_sha256"]
    assert st.verify(proof, air, cols, boundary) == g["verifies"]
    assert g["verifies"] == (g["max_degree"] < g["fri_domain"] // g["expansion"])


def test_verifier_rejects(rp):
    rows = 64
    st = ss.ScaledStark(4, 2, 2, rp.m, rows)
    trace = ss.rescue_chain_trace(rp, 5, rows)
    air, cols = ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    n_random = 17 * (st.num_randomizers * rp.m + st.max_degree(air) + 1)
    proof = st.prove(trace, air, cols, boundary, ss.stream_rng(b"r", n_random))
    assert st.verify(proof, air, cols, boundary)
    assert not st.verify(proof, air, cols, ss.rescue_chain_boundary(rp, rows, (trace[-1][0] + 1) % P))   # wrong output
    bad_cols = [list(c) for c in cols]
    bad_cols[0][17] = (bad_cols[0][17] + 1) % P                       # another computation's constants
    assert not st.verify(proof, air, bad_cols, boundary)
    # a trace with one wrong row has no proof: the transition quotient is not a polynomial
    bad = [list(r) for r in trace]
    bad[30][0] = (bad[30][0] + 1) % P
    with pytest.raises(AssertionError):
        st.prove(bad, air, cols, boundary, ss.stream_rng(b"r", n_random))

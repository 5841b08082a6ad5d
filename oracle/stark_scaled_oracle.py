"""CPU restatement of Stark::prove / Stark::verify (stark.rs:223-723) for traces far
longer than the 28 rows the reference is written for.

TEST INFRASTRUCTURE ONLY: nothing in the shipped package imports this module.

What is restated.  The reference's prover is kept object for object -- the same
randomizer rows, trace polynomials, boundary quotients, transition quotients,
randomizer polynomial, weights, nonlinear combination, FRI proof and openings, in
the same order on the proof stream -- but two things the reference cannot do at
scale are replaced by forms that give the SAME polynomials:

* its O(n^2)..O(n^3) polynomial algebra (lagrange_traditional
  univariate_poly.rs:147-164, schoolbook Mul :403-420, long Div :437-484,
  compose :203-221, zeroifier_domain :254-264).  The trace domain
  omicron^0..omicron^(n-1) (stark.rs:255-257) is a geometric progression, for
  which interpolation is two power-series products (Bostan & Schost 2005, via the
  q-binomial theorem) and the zeroifier has closed-form coefficients; products
  and exact divisions run on transforms.  Interpolants, zeroifiers and exact
  quotients are unique, so these are the reference's polynomials, coefficient for
  coefficient;
* the symbolic AIR: the reference lifts the round-constant interpolants into the
  variable x of an MPolynomial (rescue_prime.rs:309-394), which for a T-row trace
  is an expression of about 30*T terms.  Here the AIR is an MPolynomial over
  [x, current registers, next registers, constant columns]; a constant column is
  a public vector of T-1 values (the round constants of every step), interpolated
  over omicron^0..omicron^(T-2) exactly as round_constants_polynomials does.
  Substituting the column polynomials for their variables gives the reference's
  constraint polynomials, so the transition polynomials, quotients, degree bounds
  and proof bytes are the same.

Pinning.  At the reference's own size (28 rows, 128-point omicron domain) this
module's proof is byte-identical to stark_prover_oracle.Stark.prove, the line-by-
line restatement of the reference's algorithm, from the same random bytes
(tests/test_scaled_stark_cpu.py).  Beyond that size nothing of the reference
exists to compare with (it asserts omicron's value, stark.rs:65): every
interpolation is checked here by evaluating back, every division by multiplying
back, and the verifier restated below accepts.  Whole-proof parity with the
reference stays unpinned for the reason given in stark_prover_oracle.py (no
reference proof bytes exist).

Polynomials in this module are coefficient lists LOWEST degree first.
"""
from __future__ import annotations

import hashlib
import os
import sys
from typing import Callable, Dict, List, Sequence, Tuple

import stark_oracle as so
import stark_prover_oracle as sp
from stark_oracle import P, fe_inverse, fe_mul, fe_pow

_TESTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
if _TESTS not in sys.path:
    sys.path.insert(0, _TESTS)
import oracle_c as oc  # noqa: E402  (the C restatement: transforms and hashing at sizes Python cannot reach)

Boundary = List[Tuple[int, int, int]]


# ---------------------------------------------------------------------------
# transforms (ntt.rs:25-40 through the C restatement) and what is built on them
# ---------------------------------------------------------------------------
def _next_pow2(n: int) -> int:
    return 1 << max(0, (n - 1).bit_length())


def ntt(vals: Sequence[int], inverse: bool = False) -> List[int]:
    return oc.to_ints(oc.ntt(oc.from_ints(vals), inverse))


def coset_eval(coeffs: Sequence[int], n: int, offset: int) -> List[int]:
    """Values at offset * w_n^i, i < n (Polynomial::eval_domain on a shifted domain)."""
    assert len(coeffs) <= n
    if not coeffs:
        return [0] * n
    return oc.to_ints(oc.coset_lde(oc.from_ints(coeffs), n, offset))


def coset_interpolate(values: Sequence[int], offset: int) -> List[int]:
    """Inverse of coset_eval: n coefficients."""
    c = ntt(values, inverse=True)
    inv, w = fe_inverse(offset), 1
    for j in range(len(c)):
        c[j] = c[j] * w % P
        w = w * inv % P
    return c


def poly_mul(a: Sequence[int], b: Sequence[int]) -> List[int]:
    """NTT::multiply (ntt.rs:110-135)."""
    if not a or not b:
        return []
    m = len(a) + len(b) - 1
    if min(len(a), len(b)) <= 16:
        out = [0] * m
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    out[i + j] = (out[i + j] + x * y) % P
        return out
    n = _next_pow2(m)
    fa, fb = coset_eval(a, n, 1), coset_eval(b, n, 1)
    return ntt([x * y % P for x, y in zip(fa, fb)], inverse=True)[:m]


def poly_eval(coeffs: Sequence[int], x: int) -> int:      # univariate_poly.rs:33-41
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def poly_degree(coeffs: Sequence[int]) -> int:            # univariate_poly.rs:69-85
    for i in range(len(coeffs) - 1, -1, -1):
        if coeffs[i]:
            return i
    return 0


def poly_sub(a: Sequence[int], b: Sequence[int]) -> List[int]:
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % P for i in range(n)]


def batch_inverse(vals: Sequence[int]) -> List[int]:
    pre, acc = [], 1
    for v in vals:
        assert v != 0
        pre.append(acc)
        acc = acc * v % P
    inv = fe_inverse(acc)
    out = [0] * len(vals)
    for i in range(len(vals) - 1, -1, -1):
        out[i] = inv * pre[i] % P
        inv = inv * vals[i] % P
    return out


def divide_exact(num: Sequence[int], den: Sequence[int], offset: int) -> List[int]:
    """num / den for a division that leaves no remainder (univariate_poly.rs:437-484), by values on
    a coset that avoids den's zeros; checked by multiplying back."""
    dn, dd = poly_degree(num), poly_degree(den)
    assert dn >= dd
    n = _next_pow2(dn + 1)
    en, ed = coset_eval(num[:dn + 1], n, offset), coset_eval(den[:dd + 1], n, offset)
    q = coset_interpolate([x * y % P for x, y in zip(en, batch_inverse(ed))], offset)
    assert all(c == 0 for c in q[dn - dd + 1:]), "division left a remainder"
    q = q[:dn - dd + 1]
    back = poly_mul(q, den[:dd + 1])
    assert back == list(num[:dn + 1]), "division left a remainder"
    return q


# ---------------------------------------------------------------------------
# the geometric progression q^0, q^1, ..., the shape of every domain in stark.rs
# ---------------------------------------------------------------------------
class GeometricDomain:
    """Tables over u_i = prod_{j=1..i} (q^j - 1), i <= n:
    inv_u[i] = 1/u_i, e[i] = (-1)^i q^(i(i-1)/2) / u_i, qc_inv[i] = q^(-i(i-1)/2).
    By the q-binomial theorem prod_{j<k} (x - q^j) = sum_j (-1)^(k-j) q^C(k-j,2) [k over j]_q x^j with
    [k over j]_q = u_k / (u_j u_(k-j))."""

    def __init__(self, q: int, n: int):
        self.q, self.n = q, n
        u, qc, qi = [1], [1], 1
        for i in range(1, n + 1):
            qc.append(qc[-1] * qi % P)              # q^C(i,2) = q^C(i-1,2) * q^(i-1)
            qi = qi * q % P                         # q^i
            u.append(u[-1] * (qi - 1) % P)
        self.u = u
        self.inv_u = batch_inverse(u)
        self.qc_inv = batch_inverse(qc)
        self.e = [(-1 if i & 1 else 1) * qc[i] * self.inv_u[i] % P for i in range(n + 1)]

    def interpolate(self, ys: Sequence[int]) -> List[int]:
        """Polynomial::lagrange over (q^i, ys[i]), i < len(ys) (univariate_poly.rs:127-164): the Newton
        coefficients c_k satisfy ys[i]/u_i = sum_k (c_k q^C(k,2)) / u_(i-k), one product with the inverse
        series e; the change to the monomial basis, p_j u_j = sum_{k>=j} (c_k u_k) e[k-j], is a second."""
        m = len(ys)
        assert m <= self.n
        g = poly_mul([ys[i] * self.inv_u[i] % P for i in range(m)], self.e[:m])[:m]
        d = [g[k] * self.qc_inv[k] % P * self.u[k] % P for k in range(m)]
        h = poly_mul(d[::-1], self.e[:m])[:m]
        return [h[m - 1 - j] * self.inv_u[j] % P for j in range(m)]

    def lagrange_basis_at(self, m: int, x: int) -> List[int]:
        """The values l_0(x) .. l_(m-1)(x) of the Lagrange basis over q^0 .. q^(m-1) at a point x outside the
        domain: l_i(x) = Z(x) / ((x - q^i) Z'(q^i)) with Z'(q^i) = prod_{j != i} (q^i - q^j)
        = q^C(i,2) u_i * (-1)^(m-1-i) q^(i(m-1-i)) u_(m-1-i).  A second route to the interpolant's value that
        never forms its coefficients."""
        q = self.q
        q_inv2 = fe_inverse(q * q % P)
        dens, z = [], 1
        qi, qc, qf, g = 1, 1, 1, fe_pow(q, m - 2) if m >= 2 else fe_inverse(q)   # q^i, q^C(i,2), q^(i(m-1-i)), its ratio
        for i in range(m):
            d = (x - qi) % P
            assert d != 0
            z = z * d % P
            zp = qc * self.u[i] % P * qf % P * self.u[m - 1 - i] % P
            if (m - 1 - i) & 1:
                zp = P - zp
            dens.append(d * zp % P)
            qc = qc * qi % P
            qi = qi * q % P
            qf = qf * g % P                         # i(m-1-i) grows by m - 2 - 2i
            g = g * q_inv2 % P
        return [z * v % P for v in batch_inverse(dens)]

    def zeroifier(self, m: int) -> List[int]:
        """Polynomial::zeroifier_domain over q^0..q^(m-1) (univariate_poly.rs:254-264): m + 1 coefficients."""
        assert m <= self.n
        return [self.u[m] * self.e[m - j] % P * self.inv_u[j] % P for j in range(m + 1)]


# ---------------------------------------------------------------------------
# the workload: Rescue-Prime rounds chained over T rows
# ---------------------------------------------------------------------------
def rescue_chain_trace(rp: sp.RescuePrime, input_element: int, rows: int) -> List[List[int]]:
    """RescuePrime::trace (rescue_prime.rs:230-293) continued past N rounds: row r+1 is the round
    function applied to row r with the round constants of round r mod N.  rows = N + 1 is the reference's."""
    state = [input_element] + [0] * (rp.m - 1)
    out = [list(state)]
    for r in range(rows - 1):
        state = rp._round(state, r % rp.N)
        out.append(list(state))
    return out


def rescue_chain_boundary(rp: sp.RescuePrime, rows: int, output_element: int) -> Boundary:   # rescue_prime.rs:296-306
    return [(0, 1, 0), (rows - 1, 0, output_element)]


def rescue_constant_columns(rp: sp.RescuePrime, rows: int) -> List[List[int]]:
    """The 2m constant columns over the rows - 1 transitions: first-step constants of every register,
    then second-step constants (the values round_constants_polynomials interpolates, rescue_prime.rs:309-359)."""
    m, rc = rp.m, rp.round_constants
    cols = [[rc[2 * (r % rp.N) * m + i] for r in range(rows - 1)] for i in range(m)]
    cols += [[rc[2 * (r % rp.N) * m + m + i] for r in range(rows - 1)] for i in range(m)]
    return cols


def rescue_column_air(rp: sp.RescuePrime) -> List[sp.MPolynomial]:
    """transition_constraints (rescue_prime.rs:363-394) over [x, previous state, next state, first-step
    constants, second-step constants] instead of constants lifted into x."""
    m = rp.m
    v = sp.MPolynomial.variables(1 + 4 * m)
    prev, nxt, first, second = v[1:1 + m], v[1 + m:1 + 2 * m], v[1 + 2 * m:1 + 3 * m], v[1 + 3 * m:1 + 4 * m]
    air = []
    for i in range(m):
        lhs = sp.MPolynomial.constant(0)
        for k in range(m):
            lhs = lhs + sp.MPolynomial.constant(rp.MDS[i][k]) * prev[k].pow(rp.alpha)
        lhs = lhs + first[i]
        rhs = sp.MPolynomial.constant(0)
        for k in range(m):
            rhs = rhs + sp.MPolynomial.constant(rp.MDS_inv[i][k]) * (nxt[k] - second[k])
        air.append(lhs - rhs.pow(rp.alpha))
    return air


def stream_rng(seed: bytes, total: int) -> Callable[[int], bytes]:
    """Stand-in for rand::thread_rng().fill_bytes (stark.rs:244-249,345-351): consecutive slices of
    SHAKE256(seed) of `total` bytes."""
    buf = hashlib.shake_256(seed).digest(total)
    pos = {"p": 0}

    def draw(n: int) -> bytes:
        assert pos["p"] + n <= len(buf), "random bytes exhausted"
        out = buf[pos["p"]:pos["p"] + n]
        pos["p"] += n
        return out
    return draw


# ---------------------------------------------------------------------------
# stark.rs
# ---------------------------------------------------------------------------
class ScaledStark:
    def __init__(self, expansion_factor: int, num_colinearity_tests: int, security_level: int,
                 num_registers: int, original_trace_length: int):                   # stark.rs:34-86 without :65
        self.expansion_factor = expansion_factor
        self.num_colinearity_tests = num_colinearity_tests
        self.security_level = security_level
        self.num_randomizers = 4 * num_colinearity_tests
        self.num_registers = num_registers
        self.original_trace_length = original_trace_length
        randomized = original_trace_length + self.num_randomizers
        self.omicron_domain_length = 1 << (randomized * 2).bit_length()
        self.fri_domain_length = self.omicron_domain_length * expansion_factor
        self.generator = so.GENERATOR
        self.omega = so.primitive_nth_root(self.fri_domain_length)
        self.omicron = so.primitive_nth_root(self.omicron_domain_length)
        self.fri = so.Fri(self.generator, self.omega, self.fri_domain_length, expansion_factor, num_colinearity_tests)
        self.geo = GeometricDomain(self.omicron, randomized)

    # ---- boundary (stark.rs:89-131): a handful of points per register, the reference's own algebra
    def boundary_zeroifiers(self, boundary: Boundary) -> List[List[int]]:
        return [sp.zeroifier_domain([fe_pow(self.omicron, c) for (c, r, _) in boundary if r == s])[::-1]
                for s in range(self.num_registers)]

    def boundary_interpolants(self, boundary: Boundary) -> List[List[int]]:
        out = []
        for s in range(self.num_registers):
            domain = [fe_pow(self.omicron, c) for (c, r, _) in boundary if r == s]
            out.append(so.lagrange(domain, [v for (_, r, v) in boundary if r == s])[::-1])
        return out

    def transition_zeroifier(self) -> List[int]:                                    # :134-137
        return self.geo.zeroifier(self.original_trace_length - 1)

    # ---- degree bounds (:140-202); a constant column is a polynomial of degree (rows - 1) - 1
    def transition_degree_bounds(self, air: Sequence[sp.MPolynomial]) -> List[int]:
        trace_degree = self.original_trace_length + self.num_randomizers - 1
        point_degrees = [1] + [trace_degree] * (2 * self.num_registers)
        bounds = []
        for a in air:
            nv = max(len(k) for k in a.dict)
            degs = point_degrees + [self.original_trace_length - 2] * (nv - len(point_degrees))
            bounds.append(max(sum(r * l for r, l in zip(degs, k)) for k in a.dict))
        return bounds

    def transition_quotient_degree_bounds(self, air) -> List[int]:
        return [d - (self.original_trace_length - 1) for d in self.transition_degree_bounds(air)]

    def boundary_quotient_degree_bounds(self, randomized_trace_length: int, boundary: Boundary) -> List[int]:
        return [randomized_trace_length - 1 - poly_degree(bz) for bz in self.boundary_zeroifiers(boundary)]

    def max_degree(self, air) -> int:
        md = max(self.transition_quotient_degree_bounds(air))
        return 0 if md == 0 else (1 << md.bit_length()) - 1

    def constant_polynomials(self, constant_columns: Sequence[Sequence[int]]) -> List[List[int]]:
        for col in constant_columns:
            assert len(col) == self.original_trace_length - 1
        return [self.geo.interpolate(col) for col in constant_columns]

    # ---- prover (:223-471), in the steps a prover spread over several devices also runs: what happens off the
    # FRI domain is in step_*; prove() adds the codewords, commitments, FRI and openings
    def step_begin(self, trace: List[List[int]], boundary: Boundary, rng: Callable[[int], bytes]) -> dict:
        """:237-288 -> {"trace_polynomials", "boundary_quotients", "n_rows"}"""
        trace = [list(row) for row in trace]
        for _ in range(self.num_randomizers):                                       # :237-253
            trace.append([so.fe_sample(rng(17)) for _ in range(self.num_registers)])
        n_rows = len(trace)
        trace_polynomials = []
        for s in range(self.num_registers):                                         # :260-275
            column = [trace[c][s] for c in range(n_rows)]
            tp = self.geo.interpolate(column)
            full = coset_eval(tp, self.omicron_domain_length, 1)                    # evaluate back on omicron^i
            assert full[:n_rows] == column
            trace_polynomials.append(tp)
        interpolants, zeroifiers = self.boundary_interpolants(boundary), self.boundary_zeroifiers(boundary)
        # a coset for the exact divisions: any offset outside the omicron subgroup; not the FRI offset, so
        # that this restatement and the device-backed prover do not share their evaluation points
        div_offset = fe_pow(self.generator, 3)
        boundary_quotients = [divide_exact(poly_sub(trace_polynomials[s], interpolants[s]), zeroifiers[s], div_offset)
                              for s in range(self.num_registers)]                   # :282-288
        return {"trace_polynomials": trace_polynomials, "boundary_quotients": boundary_quotients, "n_rows": n_rows}

    def step_transition(self, state: dict, air: Sequence[sp.MPolynomial], constant_columns: Sequence[Sequence[int]],
                        rng: Callable[[int], bytes]) -> None:
        """:309-353 -> state["transition_quotients"], state["randomizer_poly"]"""
        trace_polynomials = state["trace_polynomials"]
        div_offset = fe_pow(self.generator, 3)
        # transition polynomials (:309-333) by values on a coset of the omicron subgroup: the composition
        # with omicron * x (:319-325) is a rotation of the values
        max_degree = self.max_degree(air)
        n_o = max(self.omicron_domain_length, max_degree + 1)      # room for every quotient coefficient
        places = n_o // self.omicron_domain_length                 # omicron = (n_o-th root)^places
        constants = self.constant_polynomials(constant_columns)
        xs = coset_eval([0, 1], n_o, div_offset)
        cur = [coset_eval(tp, n_o, div_offset) for tp in trace_polynomials]
        nxt = [c[places:] + c[:places] for c in cur]
        cons = [coset_eval(c, n_o, div_offset) for c in constants]
        columns = [xs] + cur + nxt + cons
        tz = self.transition_zeroifier()
        tz_inv = batch_inverse(coset_eval(tz, n_o, div_offset))
        tq_bounds = self.transition_quotient_degree_bounds(air)
        transition_quotients = []
        for a, bound in zip(air, tq_bounds):                                        # :336-341
            vals = _mpoly_eval_columns(a, columns, n_o)
            tq = coset_interpolate([v * z % P for v, z in zip(vals, tz_inv)], div_offset)
            assert all(c == 0 for c in tq[bound + 1:]) and tq[bound] != 0           # :379-380
            transition_quotients.append(tq[:bound + 1])
        state["transition_quotients"] = transition_quotients
        state["randomizer_poly"] = [so.fe_sample(rng(17)) for _ in range(max_degree + 1)][::-1]   # :344-353, drawn highest first

    def step_combine(self, state: dict, air: Sequence[sp.MPolynomial], boundary: Boundary, seed: bytes) -> List[int]:
        """:372-404 -> the nonlinear combination's coefficients"""
        max_degree = self.max_degree(air)
        transition_quotients, boundary_quotients = state["transition_quotients"], state["boundary_quotients"]
        tq_bounds = self.transition_quotient_degree_bounds(air)
        weights = so.sample_weights(1 + 2 * len(transition_quotients) + 2 * len(boundary_quotients), seed)   # :372-376
        terms: List[Tuple[List[int], int]] = [(state["randomizer_poly"], 0)]        # (polynomial, shift): x^shift * polynomial
        for tq, bound in zip(transition_quotients, tq_bounds):                      # :385-398
            terms += [(tq, 0), (tq, max_degree - bound)]
        bq_bounds = self.boundary_quotient_degree_bounds(state["n_rows"], boundary)
        for bq, bound in zip(boundary_quotients, bq_bounds):
            terms += [(bq, 0), (bq, max_degree - bound)]
        combination = [0] * (max_degree + 1)                                        # :401-404
        for (poly, shift), w in zip(terms, weights):
            assert len(poly) + shift <= max_degree + 1
            for j, c in enumerate(poly):
                combination[j + shift] = (combination[j + shift] + w * c) % P
        return combination

    def prove(self, trace: List[List[int]], air: Sequence[sp.MPolynomial], constant_columns: Sequence[Sequence[int]],
              boundary: Boundary, rng: Callable[[int], bytes]) -> bytes:
        ps = so.ProofStream()
        state = self.step_begin(trace, boundary, rng)
        boundary_quotients = state["boundary_quotients"]
        n_fri = self.fri_domain_length
        bq_codewords, bq_trees = [], []
        for s in range(self.num_registers):                                         # :293-306
            cw = coset_eval(boundary_quotients[s], n_fri, self.generator)
            tree = oc.merkle_tree_fe(oc.from_ints(cw))
            bq_codewords.append(cw)
            bq_trees.append(tree)
            ps.push(tree[-1].tobytes().hex())
        self.step_transition(state, air, constant_columns, rng)
        randomizer_codeword = coset_eval(state["randomizer_poly"], n_fri, self.generator)    # :356-360
        randomizer_tree = oc.merkle_tree_fe(oc.from_ints(randomizer_codeword))
        ps.push(randomizer_tree[-1].tobytes().hex())
        combination = self.step_combine(state, air, boundary, ps.prover_fiat_shamir(32))
        combined_codeword = coset_eval(combination, n_fri, self.generator)          # :406
        indices = sorted(_fri_prove(self.fri, combined_codeword, ps))               # :409-410
        duplicated = sorted(indices + [(i + self.expansion_factor) % n_fri for i in indices])   # :415-419
        for cw, tree in zip(bq_codewords, bq_trees):                                # :429-443
            for i in duplicated:
                ps.push(so.json_field_element(cw[i]))
                ps.push(so.json_hash_path(oc.merkle_open(tree, i)))
        for i in indices:                                                           # :449-464
            ps.push(so.json_field_element(randomizer_codeword[i]))
            ps.push(so.json_hash_path(oc.merkle_open(randomizer_tree, i)))
        return ps.serialize()                                                       # :470

    # ---- verifier (:474-723)
    def verify(self, proof: bytes, air: Sequence[sp.MPolynomial], constant_columns: Sequence[Sequence[int]],
               boundary: Boundary) -> bool:
        original_trace_length = 1 + max(c for (c, _, _) in boundary)
        randomized_trace_length = original_trace_length + self.num_randomizers
        ps = so.ProofStream.deserialize(proof)
        bq_roots = [ps.pull() for _ in range(self.num_registers)]
        randomizer_root = ps.pull()
        weights = so.sample_weights(1 + 2 * len(air) + 2 * len(self.boundary_interpolants(boundary)),
                                    ps.verifier_fiat_shamir(32))
        polynomial_values: List[Tuple[int, int]] = []
        if not self.fri.verify(ps, polynomial_values):
            return False
        polynomial_values.sort(key=lambda iv: iv[0])
        indices = [iv[0] for iv in polynomial_values]
        values = [iv[1] for iv in polynomial_values]
        n = self.fri.domain_length
        duplicated = sorted(indices + [(i + self.expansion_factor) % n for i in indices])
        leafs: List[Dict[int, int]] = []
        for r in range(len(bq_roots)):                                              # :532-559
            m: Dict[int, int] = {}
            for i in duplicated:
                leaf = sp._json_field_element_parse(ps.pull())
                m[i] = leaf
                path = so.json_hash_path_parse(ps.pull())
                if not so.merkle_verify(bytes.fromhex(bq_roots[r]), i, path, so.bincode_field_element(leaf)):
                    return False
            leafs.append(m)
        randomizer: Dict[int, int] = {}
        for i in indices:                                                           # :562-586
            leaf = sp._json_field_element_parse(ps.pull())
            randomizer[i] = leaf
            path = so.json_hash_path_parse(ps.pull())
            if not so.merkle_verify(bytes.fromhex(randomizer_root), i, path, so.bincode_field_element(leaf)):
                return False
        zeroifiers, interpolants = self.boundary_zeroifiers(boundary), self.boundary_interpolants(boundary)
        steps = self.original_trace_length - 1
        for col in constant_columns:
            assert len(col) == steps
        max_degree = self.max_degree(air)
        tq_bounds = self.transition_quotient_degree_bounds(air)
        bq_bounds = self.boundary_quotient_degree_bounds(randomized_trace_length, boundary)
        for i, current_index in enumerate(indices):                                 # :589-719
            x = fe_mul(self.generator, fe_pow(self.omega, current_index))
            next_index = (current_index + self.expansion_factor) % n
            x_next = fe_mul(x, self.omicron)
            current_trace, next_trace = [], []
            for s in range(self.num_registers):
                current_trace.append((leafs[s][current_index] * poly_eval(zeroifiers[s], x)
                                      + poly_eval(interpolants[s], x)) % P)
                next_trace.append((leafs[s][next_index] * poly_eval(zeroifiers[s], x_next)
                                   + poly_eval(interpolants[s], x_next)) % P)
            # the constant columns' interpolants at x through the Lagrange basis (no coefficients formed)
            basis = self.geo.lagrange_basis_at(steps, x)
            point = [x] + current_trace + next_trace + [sum(b * v for b, v in zip(basis, col)) % P
                                                        for col in constant_columns]
            # the transition zeroifier at x as the product it is (:666), one factor per trace point
            tz_x, w = 1, 1
            for _ in range(self.original_trace_length - 1):
                tz_x = tz_x * (x - w) % P
                w = w * self.omicron % P
            tz_inv = fe_inverse(tz_x)
            terms = [randomizer[current_index]]
            for s, a in enumerate(air):
                q = a.eval(point) * tz_inv % P
                terms += [q, q * fe_pow(x, max_degree - tq_bounds[s]) % P]
            for s in range(self.num_registers):
                bqv = leafs[s][current_index]
                terms += [bqv, bqv * fe_pow(x, max_degree - bq_bounds[s]) % P]
            combination = 0
            for t, w in zip(terms, weights):
                combination = (combination + t * w) % P
            if combination != values[i]:
                return False
        return True


def _mpoly_eval_columns(a: sp.MPolynomial, columns: Sequence[Sequence[int]], n: int) -> List[int]:
    """MPolynomial::eval (multivariate_poly.rs:42-72) at the n points whose coordinates are the columns."""
    out = [0] * n
    for key, coeff in a.dict.items():
        if coeff == 0:
            continue
        used = [(columns[v], e) for v, e in enumerate(key) if e]
        for i in range(n):
            t = coeff
            for col, e in used:
                t = t * (col[i] if e == 1 else pow(col[i], e, P)) % P
            out[i] = (out[i] + t) % P
    return out


def _fri_prove(fri: so.Fri, codeword: List[int], ps: so.ProofStream) -> List[int]:
    """Fri::prove (fri.rs:212-254): the commit phase through the C restatement (same roots, challenges
    and layers as so.Fri.commit), the queries from trees built once per layer."""
    n = len(codeword)
    stream = oc.Stream()
    for o in ps.objects:
        stream.push(o)
    layers, roots, _, stream = oc.fri_commit(oc.from_ints(codeword), fri.offset, fri.omega, fri.expansion_factor,
                                             fri.num_colinearity_tests, stream)
    for r in roots:
        ps.push(r.hex())                                                            # fri.rs:120
    ps.push(so.json_field_vec(oc.to_ints(layers[-1])))                              # fri.rs:146
    t = fri.num_colinearity_tests
    top = fri.sample_indices(ps.prover_fiat_shamir(32), n // 2, layers[-1].shape[0], t)   # fri.rs:218-223
    trees = [oc.merkle_tree_fe(l) for l in layers]
    indices = list(top)
    for i in range(len(layers) - 1):                                                # fri.rs:228-242
        half = layers[i].shape[0] // 2
        cs = [idx % half for idx in indices][:t]
        a, b = cs, [c + half for c in cs]
        cur, nxt = layers[i], layers[i + 1]
        for s in range(t):                                                          # fri.rs:166-179
            ps.push(so.json_string_triple(str(oc.to_int(cur[a[s]])), str(oc.to_int(cur[b[s]])), str(oc.to_int(nxt[cs[s]]))))
        for s in range(t):                                                          # fri.rs:182-204
            ps.push(so.json_hash_path(oc.merkle_open(trees[i], a[s])))
            ps.push(so.json_hash_path(oc.merkle_open(trees[i], b[s])))
            ps.push(so.json_hash_path(oc.merkle_open(trees[i + 1], cs[s])))
        indices = a + b
    return list(top) + [i + n // 2 for i in top]                                    # fri.rs:246-253

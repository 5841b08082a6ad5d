"""CPU restatement of the caller of the hot path: the polynomial algebra, the
multivariate constraint polynomials, the Rescue-Prime AIR and Stark::prove /
Stark::verify of the reference crate.

TEST INFRASTRUCTURE ONLY, like stark_oracle.py: nothing in the shipped package
imports this module.  Plain Python integers, one function per reference
function, each citing the lines it follows.

Parity pinning: the Rescue-Prime permutation is pinned by the hash known-answer
values of rescue_prime.rs:415-422 and its parameters by the tables of
rescue_prime.rs:29-159 (tests/golden/reference_kats.json); the interpolation of
test_lift (multivariate_poly.rs:294-296) pins lagrange.  The reference draws its
randomizers from rand::thread_rng() (stark.rs:244-249,345-351), so no proof bytes
of the reference exist to compare with: whole-proof parity is unpinned, and what
is checked is that this restatement's verify() accepts its own proofs and that
the device-backed prover reproduces them byte for byte from the same random
bytes.
"""
from __future__ import annotations

import hashlib
import json
from typing import Callable, Dict, List, Sequence, Tuple

import stark_oracle as so
from stark_oracle import P, fe_add, fe_sub, fe_mul, fe_neg, fe_pow, fe_inverse, fe_div

Poly = List[int]          # coefficients highest -> lowest degree, univariate_poly.rs:27


# ---------------------------------------------------------------------------
# univariate_poly.rs: the algebra Stark::prove uses
# ---------------------------------------------------------------------------
def poly_is_zero(a: Sequence[int]) -> bool:                      # :88-94
    return all(c == 0 for c in a)


def poly_add(a: Sequence[int], b: Sequence[int]) -> Poly:         # :308-342
    mx, mn = (a, b) if len(a) > len(b) else (b, a)
    d = len(mx) - len(mn)
    return list(mx[:d]) + [fe_add(mx[d + i], mn[i]) for i in range(len(mn))]


def poly_sub(a: Sequence[int], b: Sequence[int]) -> Poly:         # :345-400
    if len(b) > len(a):
        d = len(b) - len(a)
        return [fe_sub(0, b[i]) for i in range(d)] + [fe_sub(a[i], b[d + i]) for i in range(len(a))]
    d = len(a) - len(b)
    return list(a[:d]) + [fe_sub(a[d + i], b[i]) for i in range(len(b))]


def poly_neg(a: Sequence[int]) -> Poly:                           # :423-434
    return [fe_sub(0, c) for c in a]


def poly_mul(a: Sequence[int], b: Sequence[int]) -> Poly:         # :403-420
    res = [0] * (len(a) + len(b) - 1)
    for i, c1 in enumerate(a):
        if c1 == 0:
            continue
        for j, c2 in enumerate(b):
            res[i + j] = (res[i + j] + c1 * c2) % P
    return res


def poly_scale(a: Sequence[int], f: int) -> Poly:                 # :197-200
    return [fe_mul(c, f) for c in a]


def poly_div(a: Sequence[int], b: Sequence[int]) -> Poly:         # :437-484
    """Long division exactly as written: the quotient has deg(a)-deg(b)+1
    coefficients, position i of the dividend VECTOR pairs with quotient[i], and
    the divisor's first stored coefficient is taken as its leading one."""
    if poly_is_zero(b):
        raise ZeroDivisionError("Attempted to divide by zero polynomial.")
    nd, dd = so.poly_degree(a), so.poly_degree(b)
    if dd > nd:
        return [0]
    dividend = list(a)
    qd = nd - dd
    quotient = [0] * (qd + 1)
    for i in range(qd + 1):
        if i < len(dividend) and dividend[i] != 0:
            coeff = fe_div(dividend[i], b[0])
            quotient[i] = coeff
            for j in range(len(b)):
                if i + j < len(dividend):
                    dividend[i + j] = fe_sub(dividend[i + j], fe_mul(coeff, b[j]))
    return quotient


def poly_pow(a: Sequence[int], exponent: int) -> Poly:            # :285-303
    if poly_is_zero(a):
        return [0]
    if exponent == 0:
        return [1]
    acc: Poly = [1]
    # the reference walks bit positions leading_zeros-1 .. 0 of a u128; squaring [1] is [1],
    # so for exponents below 2^64 this is the ordinary left-to-right ladder
    assert exponent < (1 << 64)
    for i in reversed(range(exponent.bit_length())):
        acc = poly_mul(acc, acc)
        if (exponent >> i) & 1:
            acc = poly_mul(acc, a)
    return acc


def poly_compose(a: Sequence[int], other: Sequence[int]) -> Poly:  # :203-221
    if poly_is_zero(a):
        return [0]
    result: Poly = [0]
    power: Poly = [1]
    for coeff in reversed(a):
        result = poly_add(result, poly_mul([coeff], power))
        power = poly_mul(power, other)
    return result


def zeroifier_domain(domain: Sequence[int]) -> Poly:             # :254-264
    acc: Poly = [1]
    for d in domain:
        acc = poly_mul(acc, poly_sub([1, 0], [d]))
    return acc


def polynomial_to_json(a: Sequence[int]) -> str:                 # :224-229 (Polynomial::save)
    """serde_json of #[derive(Serialize)] struct Polynomial { coeffs: Vec<FieldElement> }."""
    return '{"coeffs":' + so.json_field_vec(a) + "}"


def polynomial_from_json(s: str) -> Poly:                        # :232-238 (Polynomial::load)
    obj = json.loads(s)
    return so.json_field_vec_parse(json.dumps(obj["coeffs"], separators=(",", ":")))


# ---------------------------------------------------------------------------
# multivariate_poly.rs
# ---------------------------------------------------------------------------
Key = Tuple[int, ...]


class MPolynomial:
    """Dictionary from exponent vectors to coefficients (multivariate_poly.rs:24-26).
    Keys keep the length they were built with, and terms whose coefficient cancels
    stay in the dictionary, both as in the reference: the degree bounds of
    stark.rs:140-168 run over every stored term."""

    def __init__(self, d: Dict[Key, int] | None = None):
        self.dict: Dict[Key, int] = dict(d or {})

    @staticmethod
    def zero() -> "MPolynomial":
        return MPolynomial()

    @staticmethod
    def constant(element: int) -> "MPolynomial":                 # :33-35
        return MPolynomial({(0,): so.fe_new(element)})

    def is_zero(self) -> bool:                                    # :38
        return all(v == 0 for v in self.dict.values())

    @staticmethod
    def variables(num_variables: int) -> List["MPolynomial"]:    # :113-130
        out = []
        for i in range(num_variables):
            e = [0] * num_variables
            e[i] = 1
            out.append(MPolynomial({tuple(e): 1}))
        return out

    @staticmethod
    def lift(poly: Sequence[int], variable_index: int) -> "MPolynomial":   # :133-146
        if poly_is_zero(poly):
            return MPolynomial.zero()
        x = MPolynomial.variables(variable_index + 1)[variable_index]
        acc = MPolynomial.zero()
        for i, coeff in enumerate(reversed(poly)):
            acc = acc + MPolynomial.constant(coeff) * x.pow(i)
        return acc

    def __add__(self, rhs: "MPolynomial") -> "MPolynomial":      # :150-172
        r = dict(self.dict)
        for k, c in rhs.dict.items():
            r[k] = fe_add(r[k], c) if k in r else c
        return MPolynomial(r)

    def __neg__(self) -> "MPolynomial":                          # :184-197
        return MPolynomial({k: fe_neg(c) for k, c in self.dict.items()})

    def __sub__(self, rhs: "MPolynomial") -> "MPolynomial":      # :175-181
        return self + (-rhs)

    def __mul__(self, rhs: "MPolynomial") -> "MPolynomial":      # :200-237
        r: Dict[Key, int] = {}
        for kl, cl in self.dict.items():
            for kr, cr in rhs.dict.items():
                e = list(kl)
                for idx, x in enumerate(kr):
                    if idx < len(e):
                        e[idx] += x
                    else:
                        e.append(x)
                k = tuple(e)
                c = fe_mul(cl, cr)
                r[k] = fe_add(r[k], c) if k in r else c
        return MPolynomial(r)

    def pow(self, exponent: int) -> "MPolynomial":               # :91-108
        if self.is_zero():
            return MPolynomial.zero()
        if exponent == 0:
            return MPolynomial.constant(1)
        # the reference takes the length of whichever key its hash map yields first; the
        # choice only pads keys with zeros and no value computed from them depends on it
        nv = len(next(iter(self.dict)))
        acc = MPolynomial({tuple([0] * nv): 1})
        for b in bin(exponent)[2:]:
            acc = acc * acc
            if b == "1":
                acc = acc * self
        return acc

    def eval(self, point: Sequence[int]) -> int:                 # :42-72
        acc = 0
        for k, c in self.dict.items():
            prod = c
            for i, e in enumerate(k):
                if i >= len(point):
                    continue
                prod = fe_mul(prod, fe_pow(point[i], e) if e else 1)
            acc = fe_add(acc, prod)
        return acc

    def eval_symbolic(self, point: Sequence[Sequence[int]]) -> Poly:    # :75-88
        acc: Poly = [0]
        for k, v in self.dict.items():
            prod: Poly = [v]
            for i in range(len(k)):
                prod = poly_mul(prod, poly_pow(point[i], k[i]))
            acc = poly_add(acc, prod)
        return acc


# ---------------------------------------------------------------------------
# rescue_prime.rs
# ---------------------------------------------------------------------------
class RescuePrime:
    """Parameters of rescue_prime.rs:27-169.  The 108 round constants are the
    Rescue-Prime specification's: SHAKE256 of "Rescue-XLIX(p,m,capacity,128)"
    cut into 17-byte little-endian integers reduced mod p (checked against the
    reference's table in the tests)."""

    def __init__(self):
        self.p = P
        self.m = 2
        self.rate = 1
        self.capacity = 1
        self.N = 27
        self.alpha = 3
        self.alpha_inv = pow(3, -1, P - 1)
        self.MDS = [[P - 3, 4], [P - 12, 13]]
        inv9 = fe_inverse(9)
        self.MDS_inv = [[fe_mul(13, inv9), fe_mul(P - 4, inv9)], [fe_mul(12, inv9), fe_mul(P - 3, inv9)]]
        bytes_per_int = (P.bit_length() + 7) // 8 + 1
        n_const = 2 * self.m * self.N
        seed = "Rescue-XLIX(%i,%i,%i,%i)" % (P, self.m, self.capacity, 128)
        stream = hashlib.shake_256(seed.encode("ascii")).digest(bytes_per_int * n_const)
        self.round_constants = [
            int.from_bytes(stream[bytes_per_int * i:bytes_per_int * (i + 1)], "little") % P for i in range(n_const)]

    def _round(self, state: List[int], r: int) -> List[int]:     # :182-222
        m = self.m
        state = [fe_pow(s, self.alpha) for s in state]
        temp = [0] * m
        for i in range(m):
            for j in range(m):
                temp[i] = fe_add(temp[i], fe_mul(self.MDS[i][j], state[j]))
        state = [fe_add(temp[i], self.round_constants[2 * r * m + i]) for i in range(m)]
        state = [fe_pow(s, self.alpha_inv) for s in state]
        temp = [0] * m
        for i in range(m):
            for j in range(m):
                temp[i] = fe_add(temp[i], fe_mul(self.MDS[i][j], state[j]))
        return [fe_add(temp[i], self.round_constants[2 * r * m + m + i]) for i in range(m)]

    def hash(self, input_element: int) -> int:                   # :172-227
        state = [input_element] + [0] * (self.m - 1)
        for r in range(self.N):
            state = self._round(state, r)
        return state[0]

    def trace(self, input_element: int) -> List[List[int]]:      # :230-293
        state = [input_element] + [0] * (self.m - 1)
        out = [list(state)]
        for r in range(self.N):
            state = self._round(state, r)
            out.append(list(state))
        return out

    def boundary_constraints(self, output_element: int) -> List[Tuple[int, int, int]]:    # :296-306
        return [(0, 1, 0), (self.N, 0, output_element)]

    def round_constants_polynomials(self, omicron: int):         # :309-359
        domain = [fe_pow(omicron, r) for r in range(self.N)]
        first, second = [], []
        for i in range(self.m):
            values = [self.round_constants[2 * r * self.m + i] for r in range(self.N)]
            first.append(MPolynomial.lift(so.lagrange(domain, values), 0))
        for i in range(self.m):
            values = [self.round_constants[2 * r * self.m + self.m + i] for r in range(self.N)]
            second.append(MPolynomial.lift(so.lagrange(domain, values), 0))
        return first, second

    def transition_constraints(self, omicron: int) -> List[MPolynomial]:    # :363-394
        first, second = self.round_constants_polynomials(omicron)
        variables = MPolynomial.variables(1 + 2 * self.m)
        previous_state = variables[1:1 + self.m]
        next_state = variables[1 + self.m:1 + 2 * self.m]
        air = []
        for i in range(self.m):
            lhs = MPolynomial.constant(0)
            for k in range(self.m):
                lhs = lhs + MPolynomial.constant(self.MDS[i][k]) * previous_state[k].pow(self.alpha)
            lhs = lhs + first[i]
            rhs = MPolynomial.constant(0)
            for k in range(self.m):
                rhs = rhs + MPolynomial.constant(self.MDS_inv[i][k]) * (next_state[k] - second[k])
            rhs = rhs.pow(self.alpha)
            air.append(lhs - rhs)
        return air


# ---------------------------------------------------------------------------
# stark.rs
# ---------------------------------------------------------------------------
Boundary = List[Tuple[int, int, int]]
RandomBytes = Callable[[int], bytes]          # n -> n random bytes (rand::thread_rng().fill_bytes)


def counter_rng(seed: bytes) -> RandomBytes:
    """Deterministic stand-in for thread_rng: the k-th call returns the first n bytes of
    SHAKE256(seed || k as u64 LE)."""
    state = {"k": 0}

    def draw(n: int) -> bytes:
        out = hashlib.shake_256(seed + state["k"].to_bytes(8, "little")).digest(n)
        state["k"] += 1
        return out
    return draw


class Stark:
    def __init__(self, expansion_factor: int, num_colinearity_tests: int, security_level: int,
                 num_registers: int, original_trace_length: int, check_omicron: bool = True):   # stark.rs:34-86
        self.expansion_factor = expansion_factor
        self.num_colinearity_tests = num_colinearity_tests
        self.security_level = security_level
        self.num_randomizers = 4 * num_colinearity_tests
        self.num_registers = num_registers
        self.original_trace_length = original_trace_length
        randomized = original_trace_length + self.num_randomizers
        product = randomized * 2
        self.omicron_domain_length = 1 << product.bit_length()
        self.fri_domain_length = self.omicron_domain_length * expansion_factor
        self.generator = so.GENERATOR
        self.omega = so.primitive_nth_root(self.fri_domain_length)
        self.omicron = so.primitive_nth_root(self.omicron_domain_length)
        self.omicron_domain = [fe_pow(self.omicron, i) for i in range(self.omicron_domain_length)]
        if check_omicron:                                        # :65 (holds for the 128-point domain only)
            assert self.omicron == 65907963977709178563567092354521124432
        self.fri = so.Fri(self.generator, self.omega, self.fri_domain_length, expansion_factor, num_colinearity_tests)

    def boundary_zeroifiers(self, boundary: Boundary) -> List[Poly]:          # :89-107
        return [zeroifier_domain([fe_pow(self.omicron, c) for (c, r, _) in boundary if r == s])
                for s in range(self.num_registers)]

    def boundary_interpolants(self, boundary: Boundary) -> List[Poly]:        # :110-131
        out = []
        for s in range(self.num_registers):
            domain = [fe_pow(self.omicron, c) for (c, r, _) in boundary if r == s]
            values = [v for (_, r, v) in boundary if r == s]
            out.append(so.lagrange(domain, values))
        return out

    def transition_zeroifier(self) -> Poly:                                   # :134-137
        return zeroifier_domain(self.omicron_domain[0:self.original_trace_length - 1])

    def transition_degree_bounds(self, transition_constraints: Sequence[MPolynomial]) -> List[int]:   # :140-168
        point_degrees = [1] + [self.original_trace_length + self.num_randomizers - 1] * (2 * self.num_randomizers)
        return [max(sum(r * l for r, l in zip(point_degrees, k)) for k in a.dict) for a in transition_constraints]

    def transition_quotient_degree_bounds(self, transition_constraints) -> List[int]:                 # :171-178
        return [d - (self.original_trace_length - 1) for d in self.transition_degree_bounds(transition_constraints)]

    def boundary_quotient_degree_bounds(self, randomized_trace_length: int, boundary: Boundary) -> List[int]:   # :181-188
        return [randomized_trace_length - 1 - so.poly_degree(bz) for bz in self.boundary_zeroifiers(boundary)]

    def max_degree(self, transition_constraints) -> int:                      # :191-202
        md = max(self.transition_quotient_degree_bounds(transition_constraints))
        return 0 if md == 0 else (1 << md.bit_length()) - 1

    @staticmethod
    def sample_weights(number: int, randomness: bytes) -> List[int]:          # :205-220
        return so.sample_weights(number, randomness)

    def _eval_fri_domain(self, poly: Sequence[int], fri_domain: Sequence[int]) -> List[int]:
        # Polynomial::eval_domain on a shifted domain = Horner at every point (univariate_poly.rs:44-54)
        return [so.poly_eval(poly, x) for x in fri_domain]

    def prove(self, trace: List[List[int]], transition_constraints: Sequence[MPolynomial], boundary: Boundary,
              rng: RandomBytes) -> bytes:                                     # :223-471
        trace = [list(row) for row in trace]
        ps = so.ProofStream()
        for _ in range(self.num_randomizers):                                 # :237-253
            trace.append([so.fe_sample(rng(17)) for _ in range(self.num_registers)])
        trace_domain = [fe_pow(self.omicron, i) for i in range(len(trace))]
        trace_polynomials = [so.lagrange(trace_domain, [trace[c][s] for c in range(len(trace))])
                             for s in range(self.num_registers)]              # :260-275
        interpolants = self.boundary_interpolants(boundary)
        zeroifiers = self.boundary_zeroifiers(boundary)
        boundary_quotients = [poly_div(poly_sub(trace_polynomials[s], interpolants[s]), zeroifiers[s])
                              for s in range(self.num_registers)]             # :282-288
        fri_domain = self.fri.eval_domain()
        bq_codewords = []
        for s in range(self.num_registers):                                   # :293-306
            cw = self._eval_fri_domain(boundary_quotients[s], fri_domain)
            bq_codewords.append(cw)
            ps.push(so.merkle_commit([so.bincode_field_element(x) for x in cw]).hex())
        point: List[Poly] = [[1, 0]] + trace_polynomials                      # :309-325
        for tp in trace_polynomials:
            point.append(poly_compose(tp, [self.omicron, 0]))
        transition_polynomials = [a.eval_symbolic(point) for a in transition_constraints]      # :328-333
        tz = self.transition_zeroifier()
        transition_quotients = [poly_div(tp, tz) for tp in transition_polynomials]             # :336-341
        max_degree = self.max_degree(transition_constraints)
        randomizer_poly = [so.fe_sample(rng(17)) for _ in range(max_degree + 1)]               # :344-353
        randomizer_codeword = self._eval_fri_domain(randomizer_poly, fri_domain)
        ps.push(so.merkle_commit([so.bincode_field_element(x) for x in randomizer_codeword]).hex())   # :356-360
        weights = self.sample_weights(1 + 2 * len(transition_quotients) + 2 * len(boundary_quotients),
                                      ps.prover_fiat_shamir(32))              # :372-376
        tq_bounds = self.transition_quotient_degree_bounds(transition_constraints)
        assert [so.poly_degree(tq) for tq in transition_quotients] == tq_bounds                # :379-380
        x: Poly = [1, 0]
        terms: List[Poly] = [randomizer_poly]                                 # :385-398
        for i, tq in enumerate(transition_quotients):
            terms.append(tq)
            terms.append(poly_mul(poly_pow(x, max_degree - tq_bounds[i]), tq))
        bq_bounds = self.boundary_quotient_degree_bounds(len(trace), boundary)
        for i in range(self.num_registers):
            terms.append(boundary_quotients[i])
            terms.append(poly_mul(poly_pow(x, max_degree - bq_bounds[i]), boundary_quotients[i]))
        combination: Poly = [0]                                               # :401-404
        for i, t in enumerate(terms):
            combination = poly_add(combination, poly_mul([weights[i]], t))
        combined_codeword = self._eval_fri_domain(combination, fri_domain)    # :406
        indices = sorted(self.fri.prove(combined_codeword, ps))               # :409-410
        n = self.fri.domain_length
        duplicated = sorted(indices + [(i + self.expansion_factor) % n for i in indices])      # :415-419
        for bqc in bq_codewords:                                              # :429-443
            leaves = [so.bincode_field_element(v) for v in bqc]
            for i in duplicated:
                ps.push(so.json_field_element(bqc[i]))
                ps.push(so.json_hash_path(so.merkle_open(i, leaves)))
        leaves = [so.bincode_field_element(v) for v in randomizer_codeword]
        for i in indices:                                                     # :449-464
            ps.push(so.json_field_element(randomizer_codeword[i]))
            ps.push(so.json_hash_path(so.merkle_open(i, leaves)))
        return ps.serialize()                                                 # :470

    def verify(self, proof: bytes, transition_constraints: Sequence[MPolynomial], boundary: Boundary) -> bool:   # :474-723
        original_trace_length = 1 + max(c for (c, _, _) in boundary)
        randomized_trace_length = original_trace_length + self.num_randomizers
        ps = so.ProofStream.deserialize(proof)
        bq_roots = [ps.pull() for _ in range(self.num_registers)]
        randomizer_root = ps.pull()
        weights = self.sample_weights(
            1 + 2 * len(transition_constraints) + 2 * len(self.boundary_interpolants(boundary)),
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
        for r in range(len(bq_roots)):                                        # :532-559
            m: Dict[int, int] = {}
            for i in duplicated:
                leaf = _json_field_element_parse(ps.pull())
                m[i] = leaf
                path = so.json_hash_path_parse(ps.pull())
                if not so.merkle_verify(bytes.fromhex(bq_roots[r]), i, path, so.bincode_field_element(leaf)):
                    return False
            leafs.append(m)
        randomizer: Dict[int, int] = {}
        for i in indices:                                                     # :562-586
            leaf = _json_field_element_parse(ps.pull())
            randomizer[i] = leaf
            path = so.json_hash_path_parse(ps.pull())
            if not so.merkle_verify(bytes.fromhex(randomizer_root), i, path, so.bincode_field_element(leaf)):
                return False
        zeroifiers = self.boundary_zeroifiers(boundary)
        interpolants = self.boundary_interpolants(boundary)
        tz = self.transition_zeroifier()
        max_degree = self.max_degree(transition_constraints)
        tq_bounds = self.transition_quotient_degree_bounds(transition_constraints)
        bq_bounds = self.boundary_quotient_degree_bounds(randomized_trace_length, boundary)
        for i, current_index in enumerate(indices):                           # :589-719
            x = fe_mul(self.generator, fe_pow(self.omega, current_index))
            next_index = (current_index + self.expansion_factor) % n
            x_next = fe_mul(x, self.omicron)
            current_trace, next_trace = [], []
            for s in range(self.num_registers):
                current_trace.append(fe_add(fe_mul(leafs[s][current_index], so.poly_eval(zeroifiers[s], x)),
                                            so.poly_eval(interpolants[s], x)))
                next_trace.append(fe_add(fe_mul(leafs[s][next_index], so.poly_eval(zeroifiers[s], x_next)),
                                         so.poly_eval(interpolants[s], x_next)))
            point = [x] + current_trace + next_trace
            tc_values = [tc.eval(point) for tc in transition_constraints]
            terms = [randomizer[current_index]]
            for s, tcv in enumerate(tc_values):
                q = fe_div(tcv, so.poly_eval(tz, x))
                terms.append(q)
                terms.append(fe_mul(q, fe_pow(x, max_degree - tq_bounds[s])))
            for s in range(self.num_registers):
                bqv = leafs[s][current_index]
                terms.append(bqv)
                terms.append(fe_mul(bqv, fe_pow(x, max_degree - bq_bounds[s])))
            combination = 0
            for j, t in enumerate(terms):
                combination = fe_add(combination, fe_mul(t, weights[j]))
            if combination != values[i]:
                return False
        return True


def _json_field_element_parse(s: str) -> int:
    """serde_json form of a FieldElement, {"value":[sign,[u32 digits...]]} (stark.rs:538)."""
    sign, digits = json.loads(s)["value"]
    v = 0
    for k, d in enumerate(digits):
        v |= d << (32 * k)
    if sign == 0:
        return 0
    return v if sign > 0 else (-v) % P

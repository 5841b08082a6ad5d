"""CPU restatement (Python big-int + hashlib) of the reference prover hot path.

TEST INFRASTRUCTURE ONLY.  This module is the parity checker for the HIP
library: it may be imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py, never by the product package.

Every function restates the algorithm of the reference function it cites
(paths are relative to the reference checkout).  Hashes come from hashlib
(BLAKE2b: RFC 7693, SHAKE256: FIPS 202), which are independent of the C
restatement in stark_oracle.c, so agreement between the two pins the C hash
code as well.

Pinning status: the field core is pinned by the reference's own known-answer
values (Rescue-Prime hash vectors rescue_prime.rs:415-422, the 128-th root
stark.rs:65, the Lagrange coefficients multivariate_poly.rs:294-296, the small
field identities field.rs:163-209,250-257).  The NTT is a mathematical
function of (p, g) and is pinned through them.  The byte encodings
(bincode/num-bigint/serde_json), Merkle roots, Fiat-Shamir bytes and FRI
codewords have no golden value anywhere in the reference and the reference
cannot be compiled here (no Rust toolchain): for those, PARITY IS UNPINNED and
rests on the crates' documented wire formats.
"""
from __future__ import annotations

import hashlib
import json
from typing import List, Sequence, Tuple

# ----------------------------------------------------------------------------
# field.rs
# ----------------------------------------------------------------------------
P = 407 * (1 << 119) + 1                      # field.rs:32
GENERATOR = 85408008396924667383611388730472331217   # field.rs:29


def fe_new(v: int) -> int:
    """field.rs:18-26 -- reduce into [0, p), negatives wrapped."""
    r = v % P        # Python % is already non-negative for positive modulus
    return r


def fe_add(a: int, b: int) -> int:            # field.rs:120-123
    return (a + b) % P


def fe_sub(a: int, b: int) -> int:            # field.rs:125-132
    return (a - b) % P


def fe_mul(a: int, b: int) -> int:            # field.rs:134-137
    return (a * b) % P


def fe_neg(a: int) -> int:                    # field.rs:144-147
    return (-a) % P


def fe_pow(base: int, exp: int) -> int:
    """field.rs:42-56 -- right-to-left square and multiply."""
    result = 1
    while exp != 0:
        if exp & 1:
            result = fe_mul(result, base)
        base = fe_mul(base, base)
        exp >>= 1
    return result


def fe_inverse(a: int) -> int:
    """field.rs:67-93 -- extended Euclid.  inverse(0) returns 0 as the
    reference's loop does (r=p, new_r=0: no iterations, t=0)."""
    t, new_t = 0, 1
    r, new_r = P, a
    while new_r != 0:
        q = r // new_r
        t, new_t = new_t, t - q * new_t
        r, new_r = new_r, r - q * new_r
    if t < 0:
        t += P
    return fe_new(t)


def fe_div(a: int, b: int) -> int:            # field.rs:139-142
    return fe_mul(a, fe_inverse(b))


def primitive_nth_root(n: int) -> int:
    """field.rs:96-107 -- square the order-2^119 generator down to order n."""
    assert n <= (1 << 119) and (n & (n - 1)) == 0, \
        "Field does not have nth root of unity where n > 2^119 or not power of two."
    root = GENERATOR
    order = 1 << 119
    while order != n:
        root = fe_pow(root, 2)
        order //= 2
    return root


def fe_sample(byte_array: bytes) -> int:
    """field.rs:110-116 -- big-endian accumulate, then mod p."""
    acc = 0
    for b in byte_array:
        acc = (acc << 8) ^ b
    return fe_new(acc % P)


# ----------------------------------------------------------------------------
# ntt.rs
# ----------------------------------------------------------------------------
class NTT:
    """ntt.rs:6-135."""

    def __init__(self, n: int):
        assert n > 0 and (n & (n - 1)) == 0, "NTT size must be power of 2"   # ntt.rs:15
        assert n <= (1 << 30), "NTT size too large for practical use"       # ntt.rs:16
        self.n = n
        self.omega = primitive_nth_root(n)
        self.omega_inv = fe_inverse(self.omega)

    def forward(self, a: List[int]) -> None:          # ntt.rs:25-28
        assert len(a) == self.n, "Input size must match NTT size"
        self._ntt(a, False)

    def inverse(self, a: List[int]) -> None:          # ntt.rs:31-40
        assert len(a) == self.n, "Input size must match NTT size"
        self._ntt(a, True)
        n_inv = fe_inverse(fe_new(self.n))
        for i in range(len(a)):
            a[i] = fe_mul(a[i], n_inv)

    def _ntt(self, a: List[int], inverse: bool) -> None:
        """ntt.rs:43-71 -- bit reversal, then iterative radix-2 DIT with a
        running-product twiddle."""
        n = len(a)
        if n <= 1:
            return
        self._bit_reverse(a)
        length = 2
        while length <= n:
            w = fe_pow(self.omega_inv if inverse else self.omega, self.n // length)
            half = length // 2
            for i in range(0, n, length):
                wn = 1
                for j in range(half):
                    u = a[i + j]
                    v = fe_mul(a[i + j + half], wn)
                    a[i + j] = fe_add(u, v)
                    a[i + j + half] = fe_sub(u, v)
                    wn = fe_mul(wn, w)
            length *= 2

    @staticmethod
    def _bit_reverse(a: List[int]) -> None:           # ntt.rs:74-88
        n = len(a)
        j = 0
        for i in range(1, n):
            bit = n >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j ^= bit
            if i < j:
                a[i], a[j] = a[j], a[i]

    def interpolate(self, evaluations: Sequence[int]) -> List[int]:   # ntt.rs:92-97
        assert len(evaluations) == self.n, "Evaluations size must match NTT size"
        c = list(evaluations)
        self.inverse(c)
        return c

    def evaluate(self, coefficients: Sequence[int]) -> List[int]:     # ntt.rs:101-107
        c = list(coefficients)[: self.n]          # Vec::resize truncates when longer
        c += [0] * (self.n - len(c))
        self.forward(c)
        return c

    def multiply(self, a: Sequence[int], b: Sequence[int]) -> List[int]:   # ntt.rs:110-135
        result_size = len(a) + len(b) - 1
        ntt_size = 1 << (result_size - 1).bit_length() if result_size > 1 else 1
        ntt = NTT(ntt_size)
        ap = list(a) + [0] * (ntt_size - len(a))
        bp = list(b) + [0] * (ntt_size - len(b))
        ntt.forward(ap)
        ntt.forward(bp)
        for i in range(ntt_size):
            ap[i] = fe_mul(ap[i], bp[i])
        ntt.inverse(ap)
        return ap[:result_size]


# ----------------------------------------------------------------------------
# univariate_poly.rs (the hot-path front door, plus the small host algebra the
# FRI verifier needs).  Coefficients are stored highest -> lowest degree
# (univariate_poly.rs:27).
# ----------------------------------------------------------------------------
def poly_eval(coeffs: Sequence[int], x: int) -> int:
    """univariate_poly.rs:33-41 -- Horner."""
    val = 0
    for c in coeffs:
        val = (val * x + c) % P
    return fe_new(val)


def is_consecutive_powers_of_unity(x: Sequence[int]) -> bool:
    """univariate_poly.rs:180-194."""
    n = len(x)
    if n <= 1 or (n & (n - 1)) != 0:
        return False
    omega = primitive_nth_root(n)
    w = 1
    for i in range(n):                     # omega.pow(i) == running product
        if x[i] != w:
            return False
        w = fe_mul(w, omega)
    return True


def poly_eval_domain_ntt(coeffs: Sequence[int], domain_size: int) -> List[int]:
    """univariate_poly.rs:57-66 -- reverse to lowest-first, pad or TRUNCATE to
    domain_size, forward NTT."""
    ntt = NTT(domain_size)
    low_first = list(reversed(coeffs))
    low_first = low_first[:domain_size] + [0] * max(0, domain_size - len(low_first))
    return ntt.evaluate(low_first)


def poly_eval_domain(coeffs: Sequence[int], domain: Sequence[int]) -> List[int]:
    """univariate_poly.rs:44-54."""
    n = len(domain)
    if n >= 8 and (n & (n - 1)) == 0 and is_consecutive_powers_of_unity(domain):
        return poly_eval_domain_ntt(coeffs, n)
    return [poly_eval(coeffs, x) for x in domain]


def poly_degree(coeffs: Sequence[int]) -> int:
    """univariate_poly.rs:69-85."""
    if not coeffs:
        return 0
    for i, c in enumerate(coeffs):
        if c != 0:
            return len(coeffs) - 1 - i
    return 0


def _poly_add(a: Sequence[int], b: Sequence[int]) -> List[int]:
    """univariate_poly.rs Add (:308-340) -- align at the low end; the result
    keeps max(len) coefficients (leading zeros are not trimmed)."""
    n = max(len(a), len(b))
    aa = [0] * (n - len(a)) + list(a)
    bb = [0] * (n - len(b)) + list(b)
    return [fe_add(x, y) for x, y in zip(aa, bb)]


def _poly_mul(a: Sequence[int], b: Sequence[int]) -> List[int]:
    """univariate_poly.rs Mul (:403-420) -- schoolbook."""
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x == 0:
            continue
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % P
    return out


def _poly_scale(a: Sequence[int], f: int) -> List[int]:    # univariate_poly.rs:197-200
    return [fe_mul(c, f) for c in a]


def lagrange_traditional(x: Sequence[int], y: Sequence[int]) -> List[int]:
    """univariate_poly.rs:104-124,147-164 -- O(n^2) basis polynomials."""
    n = len(x)
    p: List[int] = []
    for i in range(n):
        l = [1]
        for j in range(n):
            if i != j:
                num = [1, fe_neg(x[j])]
                denom = fe_inverse(fe_sub(x[i], x[j]))
                l = _poly_scale(_poly_mul(l, num), denom)
        p = _poly_add(p, _poly_scale(l, y[i]))
    return p


def lagrange_ntt(evaluations: Sequence[int]) -> List[int]:
    """univariate_poly.rs:167-177 -- inverse NTT then reverse to highest-first;
    exactly n coefficients, leading zeros kept."""
    ntt = NTT(len(evaluations))
    c = ntt.interpolate(evaluations)
    c.reverse()
    return c


def lagrange(x: Sequence[int], y: Sequence[int]) -> List[int]:
    """univariate_poly.rs:127-144."""
    assert len(x) == len(y), "x and y must be the same length"
    n = len(x)
    if n <= 8 or (n & (n - 1)) != 0:
        return lagrange_traditional(x, y)
    if is_consecutive_powers_of_unity(x):
        return lagrange_ntt(y)
    return lagrange_traditional(x, y)


def test_colinearity(points: Sequence[Tuple[int, int]]) -> bool:
    """univariate_poly.rs:267-282."""
    xs = [fe_new(px) for px, _ in points]
    ys = [fe_new(py) for _, py in points]
    return poly_degree(lagrange(xs, ys)) == 1


# ----------------------------------------------------------------------------
# Wire formats (third-party crates pinned in the reference's Cargo.lock:
# bincode 1.3.3, num-bigint 0.4.6 serde impl, serde_json 1.0.120, hex 0.4.3,
# generic-array 0.14.7 serde).  PARITY UNPINNED: documented formats, no
# reference test records a byte.
# ----------------------------------------------------------------------------
def _u32_digits(v: int) -> List[int]:
    d = []
    while v:
        d.append(v & 0xFFFFFFFF)
        v >>= 32
    return d


def bincode_field_element(v: int) -> bytes:
    """bincode::serialize(&FieldElement) as used at fri.rs:119 / stark.rs:302:
    struct{value: BigInt} -> (sign:i8, BigUint as u64-length-prefixed sequence
    of u32 LE digits, least significant first, no trailing zero digit)."""
    digits = _u32_digits(v)
    out = bytearray()
    out.append(0x00 if v == 0 else 0x01)
    out += len(digits).to_bytes(8, "little")
    for d in digits:
        out += d.to_bytes(4, "little")
    return bytes(out)


def bincode_vec_string(objs: Sequence[str]) -> bytes:
    """bincode::serialize(&Vec<String>) -- proof_stream.rs:36-38."""
    out = bytearray(len(objs).to_bytes(8, "little"))
    for s in objs:
        b = s.encode("utf-8")
        out += len(b).to_bytes(8, "little") + b
    return bytes(out)


def json_field_element(v: int) -> str:
    """serde_json::to_string(&FieldElement) -- fri.rs:146, stark.rs:433."""
    sign = 0 if v == 0 else 1
    return '{"value":[%d,[%s]]}' % (sign, ",".join(str(d) for d in _u32_digits(v)))


def json_field_vec(vs: Sequence[int]) -> str:
    return "[" + ",".join(json_field_element(v) for v in vs) + "]"


def json_field_vec_parse(s: str) -> List[int]:
    out = []
    for obj in json.loads(s):
        sign, digits = obj["value"]
        v = 0
        for i, d in enumerate(digits):
            v |= d << (32 * i)
        out.append(fe_new(-v if sign < 0 else v))
    return out


def json_hash_path(path: Sequence[bytes]) -> str:
    """serde_json of Vec<GenericArray<u8,U32>> -- fri.rs:188."""
    return "[" + ",".join("[" + ",".join(str(b) for b in h) + "]" for h in path) + "]"


def json_hash_path_parse(s: str) -> List[bytes]:
    return [bytes(h) for h in json.loads(s)]


def json_string_triple(a: str, b: str, c: str) -> str:
    """serde_json of a (String,String,String) tuple -- fri.rs:169-178."""
    return '["%s","%s","%s"]' % (a, b, c)


# ----------------------------------------------------------------------------
# merkle.rs
# ----------------------------------------------------------------------------
def _h256(data: bytes) -> bytes:
    return hashlib.blake2b(data, digest_size=32).digest()


def merkle_commit_leafs(leafs: Sequence[bytes]) -> bytes:
    """merkle.rs:17-34 -- recursive halving == level-order pairing (2i, 2i+1)."""
    assert len(leafs) > 0 and (len(leafs) & (len(leafs) - 1)) == 0, "length must be power of two"
    level = list(leafs)
    while len(level) > 1:
        level = [_h256(level[2 * i] + level[2 * i + 1]) for i in range(len(level) // 2)]
    return level[0]


def merkle_levels(data_array: Sequence[bytes]) -> List[List[bytes]]:
    """All levels of the tree, leaf digests first (what the device retains)."""
    level = [_h256(d) for d in data_array]
    assert len(level) > 0 and (len(level) & (len(level) - 1)) == 0, "length must be power of two"
    levels = [level]
    while len(level) > 1:
        level = [_h256(level[2 * i] + level[2 * i + 1]) for i in range(len(level) // 2)]
        levels.append(level)
    return levels


def merkle_commit(data_array: Sequence[bytes]) -> bytes:
    """merkle.rs:38-51."""
    return merkle_commit_leafs([_h256(d) for d in data_array])


def merkle_open(index: int, data_array: Sequence[bytes]) -> List[bytes]:
    """merkle.rs:54-93 -- path[0] is the sibling leaf digest, path[-1] the
    sibling of the root's child."""
    levels = merkle_levels(data_array)
    n = len(levels[0])
    assert index < n, "cannot open invalid index"
    path = []
    idx = index
    for lvl in levels[:-1]:
        path.append(lvl[idx ^ 1])
        idx >>= 1
    return path


def merkle_verify(root: bytes, index: int, path: Sequence[bytes], data_element: bytes) -> bool:
    """merkle.rs:96-132."""
    assert index < (1 << len(path)), "cannot verify invalid index"
    node = _h256(data_element)
    for sib in path:
        node = _h256(node + sib) if index % 2 == 0 else _h256(sib + node)
        index >>= 1
    return node == root


# ----------------------------------------------------------------------------
# proof_stream.rs
# ----------------------------------------------------------------------------
class ProofStream:
    def __init__(self):                           # proof_stream.rs:15-20
        self.objects: List[str] = []
        self.read_idx = 0

    def push(self, obj: str) -> None:             # proof_stream.rs:23-25
        self.objects.append(obj)

    def pull(self) -> str:                        # proof_stream.rs:28-33
        assert self.read_idx < len(self.objects), "ProofStream: cannot pull object; queue empty."
        obj = self.objects[self.read_idx]
        self.read_idx += 1
        return obj

    def serialize(self) -> bytes:                 # proof_stream.rs:36-38
        return bincode_vec_string(self.objects)

    @staticmethod
    def deserialize(bb: bytes) -> "ProofStream":  # proof_stream.rs:41-47
        ps = ProofStream()
        n = int.from_bytes(bb[:8], "little")
        off = 8
        for _ in range(n):
            ln = int.from_bytes(bb[off:off + 8], "little")
            off += 8
            ps.objects.append(bb[off:off + ln].decode("utf-8"))
            off += ln
        return ps

    def prover_fiat_shamir(self, num_bytes: int = 32) -> bytes:      # proof_stream.rs:50-58
        return hashlib.shake_256(self.serialize()).digest(num_bytes)

    def verifier_fiat_shamir(self, num_bytes: int = 32) -> bytes:    # proof_stream.rs:61-69
        return hashlib.shake_256(bincode_vec_string(self.objects[: self.read_idx])).digest(num_bytes)


# ----------------------------------------------------------------------------
# fri.rs
# ----------------------------------------------------------------------------
def fri_fold(codeword: Sequence[int], alpha: int, offset: int, omega: int) -> List[int]:
    """fri.rs:133-138 -- one split-and-fold step, written as the reference
    writes it (two divisions per output)."""
    n = len(codeword)
    half = n // 2
    two_inv = fe_inverse(2)
    out = []
    w = 1
    for i in range(half):
        x = fe_mul(offset, w)                      # offset * omega^i
        ax = fe_div(alpha, x)
        out.append(fe_mul(two_inv, fe_add(fe_mul(fe_add(1, ax), codeword[i]),
                                          fe_mul(fe_sub(1, ax), codeword[half + i]))))
        w = fe_mul(w, omega)
    return out


class Fri:
    def __init__(self, offset: int, omega: int, initial_domain_length: int,
                 expansion_factor: int, num_colinearity_tests: int):     # fri.rs:31-36
        self.offset = offset
        self.omega = omega
        self.domain_length = initial_domain_length
        self.expansion_factor = expansion_factor
        self.num_colinearity_tests = num_colinearity_tests
        assert self.num_rounds() > 0, "Cannot do FRI with less than 1 round"

    def num_rounds(self) -> int:                  # fri.rs:39-51
        codeword_length = self.domain_length
        num_rounds = 0
        while codeword_length > self.expansion_factor and 4 * self.num_colinearity_tests < codeword_length:
            codeword_length //= 2
            num_rounds += 1
        return num_rounds

    @staticmethod
    def sample_index(byte_array: bytes, size: int) -> int:
        """fri.rs:81-87 -- `acc: usize` shifted 8 per byte: only the last 8
        bytes survive on a 64-bit target."""
        acc = 0
        for b in byte_array:
            acc = ((acc << 8) & 0xFFFFFFFFFFFFFFFF) ^ b
        return acc % size

    def sample_indices(self, seed: bytes, size: int, reduced_size: int, number: int) -> List[int]:
        """fri.rs:54-78 -- BLAKE2b-512(seed || counter_le64), reject indices
        that collide modulo reduced_size."""
        indices: List[int] = []
        reduced: List[int] = []
        counter = 0
        while len(indices) < number:
            h = hashlib.blake2b(seed + counter.to_bytes(8, "little"), digest_size=64).digest()
            index = Fri.sample_index(h, size)
            r = index % reduced_size
            counter += 1
            if r not in reduced:
                indices.append(index)
                reduced.append(r)
        return indices

    def eval_domain(self) -> List[int]:           # fri.rs:90-97
        out = []
        w = 1
        for _ in range(self.domain_length):
            out.append(fe_mul(self.offset, w))
            w = fe_mul(w, self.omega)
        return out

    def commit(self, codeword: Sequence[int], proof_stream: ProofStream) -> List[List[int]]:
        """fri.rs:100-152."""
        codeword = list(codeword)
        omega = self.omega
        offset = self.offset
        codewords: List[List[int]] = []
        rounds = self.num_rounds()
        for r in range(rounds):
            n = len(codeword)
            assert fe_pow(omega, n - 1) == fe_inverse(omega), \
                "error in commit: omega does not have the right order!"
            root = merkle_commit([bincode_field_element(x) for x in codeword])
            proof_stream.push(root.hex())
            if r == rounds - 1:
                break
            alpha = fe_sample(proof_stream.prover_fiat_shamir(32))
            codewords.append(list(codeword))
            codeword = fri_fold(codeword, alpha, offset, omega)
            omega = fe_pow(omega, 2)
            offset = fe_pow(offset, 2)
        proof_stream.push(json_field_vec(codeword))
        codewords.append(list(codeword))
        return codewords

    def query(self, current_codeword: Sequence[int], next_codeword: Sequence[int],
              c_indices: Sequence[int], proof_stream: ProofStream) -> List[int]:
        """fri.rs:155-209."""
        a_indices = list(c_indices)
        b_indices = [i + len(current_codeword) // 2 for i in c_indices]
        for s in range(self.num_colinearity_tests):
            proof_stream.push(json_string_triple(str(current_codeword[a_indices[s]]),
                                                 str(current_codeword[b_indices[s]]),
                                                 str(next_codeword[c_indices[s]])))
        cur_ser = [bincode_field_element(x) for x in current_codeword]
        nxt_ser = [bincode_field_element(x) for x in next_codeword]
        cur_levels = merkle_levels(cur_ser)
        nxt_levels = merkle_levels(nxt_ser)

        def _open(levels, index):
            assert index < len(levels[0]), "cannot open invalid index"
            path, idx = [], index
            for lvl in levels[:-1]:
                path.append(lvl[idx ^ 1])
                idx >>= 1
            return path

        for s in range(self.num_colinearity_tests):
            proof_stream.push(json_hash_path(_open(cur_levels, a_indices[s])))
            proof_stream.push(json_hash_path(_open(cur_levels, b_indices[s])))
            proof_stream.push(json_hash_path(_open(nxt_levels, c_indices[s])))
        return a_indices + b_indices

    def prove(self, codeword: Sequence[int], proof_stream: ProofStream) -> List[int]:
        """fri.rs:212-254."""
        codewords = self.commit(codeword, proof_stream)
        top_level_indices = self.sample_indices(proof_stream.prover_fiat_shamir(32),
                                                len(codewords[0]) // 2, len(codewords[-1]),
                                                self.num_colinearity_tests)
        indices = list(top_level_indices)
        for i in range(len(codewords) - 1):
            indices = [idx % (len(codewords[i]) // 2) for idx in indices]
            indices = self.query(codewords[i], codewords[i + 1], indices, proof_stream)
        return list(top_level_indices) + [i + len(codewords[0]) // 2 for i in top_level_indices]

    def verify(self, proof_stream: ProofStream, polynomial_values: List[Tuple[int, int]]) -> bool:
        """fri.rs:256-417."""
        omega = self.omega
        offset = self.offset
        rounds = self.num_rounds()
        roots: List[str] = []
        alphas: List[int] = []
        for _ in range(rounds):
            roots.append(proof_stream.pull())
            alphas.append(fe_sample(proof_stream.verifier_fiat_shamir(32)))
        last_codeword = json_field_vec_parse(proof_stream.pull())
        root = merkle_commit([bincode_field_element(x) for x in last_codeword]).hex()
        if roots[-1] != root:
            return False
        degree = (len(last_codeword) // self.expansion_factor) - 1
        last_omega, last_offset = omega, offset
        for _ in range(rounds - 1):
            last_omega = fe_pow(last_omega, 2)
            last_offset = fe_pow(last_offset, 2)
        assert fe_inverse(last_omega) == fe_pow(last_omega, len(last_codeword) - 1)
        last_domain = [fe_mul(last_offset, fe_pow(last_omega, i)) for i in range(len(last_codeword))]
        poly = lagrange(last_domain, last_codeword)
        assert poly_eval_domain(poly, last_domain) == last_codeword, "last codeword is not low degree"
        if poly_degree(poly) > degree:
            return False
        top_level_indices = self.sample_indices(proof_stream.verifier_fiat_shamir(32),
                                                self.domain_length >> 1,
                                                self.domain_length >> (rounds - 1),
                                                self.num_colinearity_tests)
        for r in range(rounds - 1):
            c_indices = [idx % (self.domain_length >> (r + 1)) for idx in top_level_indices]
            a_indices = list(c_indices)
            b_indices = [idx + (self.domain_length >> (r + 1)) for idx in c_indices]
            aa, bb, cc = [], [], []
            for s in range(self.num_colinearity_tests):
                ay_s, by_s, cy_s = json.loads(proof_stream.pull())
                ay, by, cy = fe_new(int(ay_s)), fe_new(int(by_s)), fe_new(int(cy_s))
                aa.append(ay); bb.append(by); cc.append(cy)
                if r == 0:
                    polynomial_values.append((a_indices[s], ay))
                    polynomial_values.append((b_indices[s], by))
                ax = fe_mul(offset, fe_pow(omega, a_indices[s]))
                bx = fe_mul(offset, fe_pow(omega, b_indices[s]))
                cx = alphas[r]
                if not test_colinearity([(ax, ay), (bx, by), (cx, cy)]):
                    return False
            for i in range(self.num_colinearity_tests):
                path = json_hash_path_parse(proof_stream.pull())
                if not merkle_verify(bytes.fromhex(roots[r]), a_indices[i], path,
                                     bincode_field_element(aa[i])):
                    return False
                path = json_hash_path_parse(proof_stream.pull())
                if not merkle_verify(bytes.fromhex(roots[r]), b_indices[i], path,
                                     bincode_field_element(bb[i])):
                    return False
                path = json_hash_path_parse(proof_stream.pull())
                if not merkle_verify(bytes.fromhex(roots[r + 1]), c_indices[i], path,
                                     bincode_field_element(cc[i])):
                    return False
            omega = fe_pow(omega, 2)
            offset = fe_pow(offset, 2)
        return True


# ----------------------------------------------------------------------------
# stark.rs:205-220 -- weights from the transcript (host-side helper)
# ----------------------------------------------------------------------------
def sample_weights(number: int, randomness: bytes) -> List[int]:
    out = []
    for i in range(number):
        h = hashlib.blake2b(randomness + i.to_bytes(8, "little"), digest_size=32).digest()
        out.append(fe_sample(h))
    return out


# ----------------------------------------------------------------------------
# Coset low-degree extension.  The reference evaluates on a shifted domain by
# Horner at every point (univariate_poly.rs:44-54 falls to :53 because the
# domain is not omega^i); the value is sum_j c_j (offset*omega^i)^j, which
# equals the size-n NTT of c_j*offset^j.  Both forms are given so tests can
# check one against the other at small sizes.
# ----------------------------------------------------------------------------
def coset_lde_horner(coeffs_low_first: Sequence[int], n_out: int, offset: int) -> List[int]:
    omega = primitive_nth_root(n_out)
    hi_first = list(reversed(coeffs_low_first))
    out, w = [], 1
    for _ in range(n_out):
        out.append(poly_eval(hi_first, fe_mul(offset, w)))
        w = fe_mul(w, omega)
    return out


def coset_lde_ntt(coeffs_low_first: Sequence[int], n_out: int, offset: int) -> List[int]:
    assert len(coeffs_low_first) <= n_out
    shifted, s = [], 1
    for c in coeffs_low_first:
        shifted.append(fe_mul(c, s))
        s = fe_mul(s, offset)
    return NTT(n_out).evaluate(shifted)


# ----------------------------------------------------------------------------
# Packed element helpers shared by the tests: 16 bytes, two little-endian u64.
# ----------------------------------------------------------------------------
def pack(vals: Sequence[int]) -> bytes:
    return b"".join(int(v).to_bytes(16, "little") for v in vals)


def unpack(buf: bytes) -> List[int]:
    return [int.from_bytes(buf[i:i + 16], "little") for i in range(0, len(buf), 16)]


def splitmix_field_vector(n: int, seed: int = 0x5741524B) -> List[int]:
    """Deterministic synthetic field vector: SplitMix64 stream, two words per
    candidate, rejection-sampled below p.  The C oracle and the benchmark use
    the same generator so data can be regenerated anywhere."""
    mask = 0xFFFFFFFFFFFFFFFF
    state = seed & mask
    out = []

    def nxt():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        return z ^ (z >> 31)

    while len(out) < n:
        lo = nxt()
        hi = nxt()
        v = (hi << 64) | lo
        if v < P:
            out.append(v)
    return out

"""Pins the two CPU restatements (oracle/stark_oracle.py, oracle/stark_oracle.c):
 * against the reference's own known-answer data (tests/golden/reference_kats.json),
 * against independent standard implementations (hashlib BLAKE2b / SHAKE256),
 * against each other, and against the committed vectors.
No GPU needed."""
import hashlib

import numpy as np
import pytest

import oracle_c as oc
import stark_oracle as so

P = so.P


def _ev(s):
    return eval(s, {"p": P}) % P


# --- reference known answers -------------------------------------------------
class _PyF:
    mul = staticmethod(so.fe_mul)
    add = staticmethod(so.fe_add)
    pow = staticmethod(so.fe_pow)


class _CF:
    @staticmethod
    def mul(a, b):
        o = np.zeros(2, dtype=np.uint64)
        oc.lib.so_fe_mul(oc.p64(oc.fe(a)), oc.p64(oc.fe(b)), oc.p64(o))
        return oc.to_int(o)

    @staticmethod
    def add(a, b):
        o = np.zeros(2, dtype=np.uint64)
        oc.lib.so_fe_add(oc.p64(oc.fe(a)), oc.p64(oc.fe(b)), oc.p64(o))
        return oc.to_int(o)

    @staticmethod
    def pow(a, e):
        o = np.zeros(2, dtype=np.uint64)
        oc.lib.so_fe_pow(oc.p64(oc.fe(a)), oc.p64(oc.fe(e)), oc.p64(o))
        return oc.to_int(o)


def _rescue_hash(F, params, x):
    """rescue_prime.rs:172-227 with the reference's parameters (fixture data)."""
    m, N = params["m"], params["N"]
    alpha, alpha_inv = params["alpha"], int(params["alpha_inv"])
    mds = [int(v) for v in params["MDS"]]
    rc = [int(v) for v in params["round_constants"]]
    state = [x] + [0] * (m - 1)
    for r in range(N):
        for half, e in ((0, alpha), (1, alpha_inv)):
            state = [F.pow(s, e) for s in state]
            temp = [0] * m
            for i in range(m):
                for j in range(m):
                    temp[i] = F.add(temp[i], F.mul(mds[i * m + j], state[j]))
            state = [F.add(temp[i], rc[2 * r * m + half * m + i]) for i in range(m)]
    return state[0]


@pytest.mark.parametrize("F", [_PyF, _CF], ids=["python", "c"])
def test_rescue_prime_kats(F, ref_kats):
    rp = ref_kats["rescue_prime"]
    for inp, outp in rp["hash_kats"]:
        assert _rescue_hash(F, rp, int(inp)) == int(outp)


def test_reference_scalars(ref_kats):
    assert so.GENERATOR == int(ref_kats["generator"])
    assert so.primitive_nth_root(128) == int(ref_kats["omicron_128th_root"])
    o = np.zeros(2, dtype=np.uint64)
    assert oc.lib.so_primitive_nth_root(128, oc.p64(o)) == 0
    assert oc.to_int(o) == int(ref_kats["omicron_128th_root"])
    g = np.zeros(2, dtype=np.uint64)
    oc.lib.so_generator(oc.p64(g))
    assert oc.to_int(g) == so.GENERATOR
    assert so.fe_pow(so.GENERATOR, 1 << 119) == 1 and so.fe_pow(so.GENERATOR, 1 << 118) != 1
    for ident in ref_kats["field_identities"]:
        a, b, out = _ev(ident["a"]), _ev(ident["b"]), _ev(ident["out"])
        for F in (_PyF, _CF):
            if ident["op"] == "add":
                assert F.add(a, b) == out
            elif ident["op"] == "pow":
                assert F.pow(a, b) == out
        if ident["op"] == "sub":
            assert so.fe_sub(a, b) == out
            o = np.zeros(2, dtype=np.uint64)
            oc.lib.so_fe_sub(oc.p64(oc.fe(a)), oc.p64(oc.fe(b)), oc.p64(o))
            assert oc.to_int(o) == out


def test_lagrange_lift_kat(ref_kats):
    k = ref_kats["lagrange_lift"]
    poly = so.lagrange([int(v) for v in k["x"]], [int(v) for v in k["y"]])
    assert list(reversed(poly)) == [int(v) for v in k["coeffs_low_first"]]


def test_inverse_matches_euclid():
    vals = so.splitmix_field_vector(16, seed=5) + [1, 2, 3, P - 1, 0]
    for v in vals:
        o = np.zeros(2, dtype=np.uint64)
        oc.lib.so_fe_inverse(oc.p64(oc.fe(v)), oc.p64(o))
        assert oc.to_int(o) == so.fe_inverse(v)
        if v:
            assert so.fe_mul(v, so.fe_inverse(v)) == 1


# --- hashes against hashlib ----------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 9, 25, 64, 127, 128, 129, 255, 256, 257, 1000])
def test_blake2b_vs_hashlib(n):
    data = hashlib.shake_256(b"b2" + bytes([n % 256])).digest(n)
    assert oc.blake2b(data, 32) == hashlib.blake2b(data, digest_size=32).digest()
    assert oc.blake2b(data, 64) == hashlib.blake2b(data, digest_size=64).digest()


@pytest.mark.parametrize("n", [0, 1, 135, 136, 137, 272, 1000])
def test_shake256_vs_hashlib(n):
    data = hashlib.sha512(bytes([n % 256])).digest() * 20
    data = data[:n]
    for outlen in (32, 136, 200):
        assert oc.shake256(data, outlen) == hashlib.shake_256(data).digest(outlen)


# --- C vs Python restatement ---------------------------------------------------
def test_field_vectors(golden):
    f = golden["field"]
    a = so.unpack(bytes.fromhex(f["a"]))
    b = so.unpack(bytes.fromhex(f["b"]))
    assert so.unpack(bytes.fromhex(f["mul"])) == [so.fe_mul(x, y) for x, y in zip(a, b)]
    aa, bb = oc.from_ints(a), oc.from_ints(b)
    out = np.empty_like(aa)
    oc.lib.so_fe_mul_vec(oc.p64(aa), oc.p64(bb), oc.p64(out), len(a))
    assert out.tobytes().hex() == f["mul"]
    for x, y, s, d in zip(a, b, so.unpack(bytes.fromhex(f["add"])), so.unpack(bytes.fromhex(f["sub"]))):
        assert _CF.add(x, y) == s == so.fe_add(x, y)
        assert so.fe_sub(x, y) == d
    for k, v in golden["roots"].items():
        o = np.zeros(2, dtype=np.uint64)
        oc.lib.so_primitive_nth_root(1 << int(k), oc.p64(o))
        assert oc.to_int(o) == int(v) == so.primitive_nth_root(1 << int(k))
    for hx, v in golden["sample"]:
        s = np.frombuffer(bytes.fromhex(hx), dtype=np.uint8).copy()
        o = np.zeros(2, dtype=np.uint64)
        oc.lib.so_fe_sample(oc.p8(s), len(s), oc.p64(o))
        assert oc.to_int(o) == int(v) == so.fe_sample(bytes.fromhex(hx))


def test_mul_random_vs_bigint():
    rng = np.random.default_rng(1)
    vals = [int.from_bytes(rng.bytes(16), "little") % P for _ in range(2000)]
    a, b = vals[:1000], vals[1000:]
    out = np.empty((1000, 2), dtype=np.uint64)
    oc.lib.so_fe_mul_vec(oc.p64(oc.from_ints(a)), oc.p64(oc.from_ints(b)), oc.p64(out), 1000)
    assert oc.to_ints(out) == [(x * y) % P for x, y in zip(a, b)]


@pytest.mark.parametrize("n", [8, 64, 1024])
def test_ntt_golden(n, golden):
    g = golden["ntt"][str(n)]
    x = so.splitmix_field_vector(n, seed=g["seed"])
    assert oc.to_ints(oc.splitmix(n, g["seed"])) == x
    xa = oc.from_ints(x)
    assert oc.ntt(xa).tobytes().hex() == g["forward"]
    assert oc.ntt(xa, inverse=True).tobytes().hex() == g["inverse"]
    f = list(x)
    so.NTT(n).forward(f)
    assert so.pack(f).hex() == g["forward"]
    # definition check: X[k] = sum_j x[j] w^(jk)
    w = so.primitive_nth_root(n)
    for k in (0, 1, n - 1):
        assert f[k] == sum(x[j] * pow(w, j * k, P) for j in range(n)) % P
    # round trip
    assert np.array_equal(oc.ntt(oc.ntt(xa), inverse=True), xa)


def test_ntt_u64_inputs_golden(golden):
    g = golden["ntt"]["4096_u64"]
    x = oc.splitmix(4096, g["seed"])
    x[:, 1] = 0
    assert hashlib.sha256(oc.ntt(x).tobytes()).hexdigest() == g["forward_sha256"]


def test_ntt_reference_unit_cases():
    """ntt.rs:143-206 restated: round trips at n=4, 8 and (1+x)(1+2x)."""
    for n, coeffs in ((4, [1, 2, 3, 4]), (8, [i * i + 1 for i in range(8)])):
        ntt = so.NTT(n)
        assert ntt.interpolate(ntt.evaluate(coeffs)) == coeffs
        assert ntt.evaluate(ntt.interpolate(coeffs)) == coeffs
    assert so.NTT(8).multiply([1, 1], [1, 2]) == [1, 3, 2]
    with pytest.raises(AssertionError, match="power of 2"):
        so.NTT(12)
    with pytest.raises(AssertionError, match="must match"):
        so.NTT(8).forward([1, 2, 3])


def test_eval_domain_and_lagrange_equivalence():
    """univariate_poly.rs:820-868 restated: NTT path == Horner / Lagrange."""
    for n in (16, 32):
        w = so.primitive_nth_root(n)
        domain = [so.fe_pow(w, i) for i in range(n)]
        coeffs = so.splitmix_field_vector(n, seed=n)           # highest-first
        assert so.poly_eval_domain(coeffs, domain) == [so.poly_eval(coeffs, x) for x in domain]
        ys = so.splitmix_field_vector(n, seed=n + 1)
        assert so.lagrange(domain, ys) == so.lagrange_traditional(domain, ys)
    # truncation semantics (ntt.rs:104): size+1 coefficients on size points drop the top degree
    n = 16
    w = so.primitive_nth_root(n)
    domain = [so.fe_pow(w, i) for i in range(n)]
    coeffs = so.splitmix_field_vector(n + 1, seed=3)
    got = so.poly_eval_domain(coeffs, domain)
    assert got == [so.poly_eval(coeffs[1:], x) for x in domain]
    # shifted domain falls back to Horner
    shifted = [so.fe_mul(so.GENERATOR, x) for x in domain]
    assert not so.is_consecutive_powers_of_unity(shifted)


def test_coset_lde(golden):
    g = golden["coset_lde"]
    c = so.splitmix_field_vector(g["n_coeffs"], seed=g["seed"])
    ca = oc.from_ints(c)
    out = oc.coset_lde(ca, g["n_out"], int(g["offset"]))
    assert out.tobytes().hex() == g["out"]
    h = np.empty_like(out)
    oc.lib.so_coset_lde_horner(oc.p64(ca), len(c), g["n_out"], oc.p64(oc.fe(int(g["offset"]))), oc.p64(h))
    assert np.array_equal(h, out)


def test_leaf_bytes(golden):
    for v, hx in golden["leaf_bytes"]:
        assert oc.bincode_fe(int(v)).hex() == hx == so.bincode_field_element(int(v)).hex()
    # documented examples of the wire format
    assert so.bincode_field_element(0).hex() == "00" + "00" * 8
    assert so.bincode_field_element(1).hex() == "01" + "0100000000000000" + "01000000"
    assert so.bincode_field_element(P - 1).hex() == "01" + "0400000000000000" + "00000000" * 3 + "000080cb"


def test_merkle(golden):
    for n, g in golden["merkle_fe"].items():
        elems = np.frombuffer(bytes.fromhex(g["elems"]), dtype=np.uint64).reshape(-1, 2).copy()
        assert oc.merkle_commit_fe(elems).hex() == g["root"]
        tree = oc.merkle_tree_fe(elems)
        assert tree[-1].tobytes().hex() == g["root"]
        path = oc.merkle_open(tree, 1)
        assert [p.hex() for p in path] == g["open_1"]
        leaves = [so.bincode_field_element(v) for v in oc.to_ints(elems)]
        assert so.merkle_commit(leaves).hex() == g["root"]
        assert so.merkle_verify(bytes.fromhex(g["root"]), 1, path, leaves[1])
        assert not so.merkle_verify(bytes.fromhex(g["root"]), 0, path, leaves[1])
    g = golden["merkle_bytes"]
    leaves = [bytes.fromhex(l) for l in g["leaves"]]
    assert oc.merkle_commit_bytes(leaves).hex() == g["root"] == so.merkle_commit(leaves).hex()
    assert [h.hex() for h in so.merkle_open(5, leaves)] == g["open_5"]


def test_merkle_reference_unit_case():
    """merkle.rs:157-206 restated (deterministic leaves instead of thread_rng)."""
    n = 64
    leaves = [hashlib.shake_256(bytes([i])).digest(1 + (i * 7) % 99) for i in range(n)]
    root = so.merkle_commit(leaves)
    fake_root = so.merkle_commit(list(reversed(leaves)))
    for i in range(n):
        path = so.merkle_open(i, leaves)
        assert so.merkle_verify(root, i, path, leaves[i])
        assert not so.merkle_verify(root, i, path, leaves[i] + b"x")
        assert not so.merkle_verify(root, (i + 1) % n, path, leaves[i])
        assert not so.merkle_verify(fake_root, i, path, leaves[i])
        for j in range(len(path)):
            bad = list(path)
            bad[j] = bytes(32)
            assert not so.merkle_verify(root, i, bad, leaves[i])
    with pytest.raises(AssertionError, match="power of two"):
        so.merkle_commit(leaves[:3])
    with pytest.raises(AssertionError, match="invalid index"):
        so.merkle_open(64, leaves)


def test_proof_stream():
    """proof_stream.rs:76-95 restated + C agreement."""
    ps = so.ProofStream()
    cs = oc.Stream()
    for s in ("1", "2", "3", "4", "a" * 200):
        ps.push(s)
        cs.push(s)
    assert cs.serialize() == ps.serialize()
    assert cs.fiat_shamir(32) == ps.prover_fiat_shamir(32)
    ps2 = so.ProofStream.deserialize(ps.serialize())
    assert [ps2.pull() for _ in range(5)] == ps.objects
    assert ps2.prover_fiat_shamir(32) == ps.prover_fiat_shamir(32)
    with pytest.raises(AssertionError, match="queue empty"):
        ps2.pull()


def test_fri_fold_forms_agree():
    n = 256
    cw = so.splitmix_field_vector(n, seed=9)
    alpha = so.fe_sample(b"\x01" * 32)
    omega = so.primitive_nth_root(n)
    lit = so.fri_fold(cw, alpha, so.GENERATOR, omega)
    fast = oc.fri_fold(oc.from_ints(cw), alpha, so.GENERATOR, omega)
    assert oc.to_ints(fast) == lit


def test_fri_commit_reference_test_params(golden):
    g = golden["fri_commit_ref_test"]
    cw = np.frombuffer(bytes.fromhex(g["codeword"]), dtype=np.uint64).reshape(-1, 2).copy()
    omega = so.primitive_nth_root(g["n0"])
    layers, roots, alphas, stream = oc.fri_commit(cw, so.GENERATOR, omega, g["expansion"], g["ncol"])
    assert stream.objects == g["objects"]
    assert [len(l) for l in layers] == g["layer_lens"]
    assert layers[-1].tobytes().hex() == g["last_codeword"]
    assert [r.hex() for r in roots] == g["objects"][:-1]


def test_fri_commit_4096(golden):
    g = golden["fri_commit_4096"]
    coeffs = oc.splitmix(g["n0"] // 4, g["seed"])
    cw = oc.coset_lde(coeffs, g["n0"], so.GENERATOR)
    omega = so.primitive_nth_root(g["n0"])
    layers, roots, alphas, stream = oc.fri_commit(cw, so.GENERATOR, omega, g["expansion"], g["ncol"])
    assert len(roots) == g["rounds"]
    assert [r.hex() for r in roots] == g["roots"]
    assert [hashlib.sha256(l.tobytes()).hexdigest() for l in layers] == g["layers_sha256"]
    assert hashlib.sha256(stream.objects[-1].encode()).hexdigest() == g["last_codeword_json_sha256"]


def test_fri_prove_verify_reference_test(golden):
    """fri.rs:428-504 restated: honest proof verifies, opened points match the
    polynomial, corrupted codeword is rejected; stream matches the vectors."""
    g = golden["fri_commit_ref_test"]
    gp = golden["fri_prove_ref_test"]
    n0 = g["n0"]
    omega = so.primitive_nth_root(n0)
    fri = so.Fri(so.GENERATOR, omega, n0, g["expansion"], g["ncol"])
    coeffs = [so.fe_new(i) for i in range(64)]
    codeword = so.unpack(bytes.fromhex(g["codeword"]))
    ps = so.ProofStream()
    idx = fri.prove(codeword, ps)
    assert idx == gp["indices"]
    assert hashlib.sha256(ps.serialize()).hexdigest() == gp["serialized_sha256"]
    points = []
    assert fri.verify(ps, points)
    for i, v in points:
        assert so.poly_eval(coeffs, so.fe_pow(omega, i)) == v
    bad = list(codeword)
    for i in range(63 // 3):
        bad[i] = 0
    ps = so.ProofStream()
    fri.prove(bad, ps)
    assert fri.verify(ps, []) is False
    # C sample_indices agrees
    seed = hashlib.sha256(b"seed").digest()
    assert oc.fri_sample_indices(seed, 128, 128, 17) == fri.sample_indices(seed, 128, 128, 17)


def test_num_rounds():
    assert so.Fri(so.GENERATOR, so.primitive_nth_root(1 << 24), 1 << 24, 4, 17).num_rounds() == 18
    assert oc.lib.so_fri_num_rounds(1 << 24, 4, 17) == 18
    assert oc.lib.so_fri_num_rounds(1 << 24, 4, 2) == 21
    assert oc.lib.so_fri_num_rounds(256, 4, 17) == 2
    assert oc.lib.so_fri_num_rounds(512, 4, 2) == 6
    with pytest.raises(AssertionError):
        so.Fri(1, 1, 4, 4, 1)


def test_threaded_loops_give_the_same_values():
    """The restatement's bulk loops on several threads (the all-cores CPU baseline) against the sequential code:
    transforms of sizes on both sides of the run length, a tree, a fold and a whole Fri::commit."""
    for log_n in (3, 11, 13, 16):
        x = oc.splitmix(1 << log_n, 900 + log_n)
        one_f, one_i = oc.ntt(x), oc.ntt(x, inverse=True)
        with oc.threads(4):
            assert np.array_equal(oc.ntt(x), one_f) and np.array_equal(oc.ntt(x, inverse=True), one_i)
    x = oc.splitmix(1 << 14, 77)
    omega = so.primitive_nth_root(1 << 14)
    tree = oc.merkle_tree_fe(x)
    fold = oc.fri_fold(x, 12345, so.GENERATOR, omega)
    layers, roots, alphas, _ = oc.fri_commit(x, so.GENERATOR, omega, 4, 17)
    with oc.threads(3):
        assert np.array_equal(oc.merkle_tree_fe(x), tree)
        assert np.array_equal(oc.fri_fold(x, 12345, so.GENERATOR, omega), fold)
        l2, r2, a2, _ = oc.fri_commit(x, so.GENERATOR, omega, 4, 17)
    assert r2 == roots and a2 == alphas and all(np.array_equal(u, v) for u, v in zip(l2, layers))
    # and the fold's runs against the literal formula of fri.rs:133-138 at a few positions
    vals = oc.to_ints(x)
    two_inv = so.fe_inverse(2)
    for i in (0, 1, 4095, 4096, 4097, (1 << 13) - 1):
        r = so.fe_div(12345, so.fe_mul(so.GENERATOR, so.fe_pow(omega, i)))
        want = so.fe_mul(two_inv, so.fe_add(so.fe_mul(so.fe_add(1, r), vals[i]),
                                            so.fe_mul(so.fe_sub(1, r), vals[(1 << 13) + i])))
        assert oc.to_int(fold[i]) == want

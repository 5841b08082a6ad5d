"""Parity of the HIP library (through the C ABI) with the CPU restatements.
Bit-exact everywhere: all results are canonical integers, digests or bytes."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_c as oc          # noqa: E402
import stark_oracle as so      # noqa: E402

P, G = so.P, so.GENERATOR


@pytest.fixture(scope="module")
def ctx():
    from stark_amd import Context
    c = Context(0)
    yield c
    c.close()


def _edge_vector(n, seed):
    x = oc.splitmix(n, seed)
    edge = [0, 1, P - 1, 2, (1 << 64) - 1, 1 << 64, (1 << 96) + 5, P - 2]
    for i, v in enumerate(edge[: max(0, min(len(edge), n - 1))]):
        x[(i * 7 + 1) % n] = oc.fe(v)
    return x


# ---- NTT (ntt.rs:25-71) ------------------------------------------------------
@pytest.mark.parametrize("log_n", list(range(0, 19)) + [20])
def test_ntt_forward_inverse_vs_oracle(ctx, log_n):
    n = 1 << log_n
    x = _edge_vector(n, 1000 + log_n)
    f = ctx.ntt(x)
    assert np.array_equal(f, oc.ntt(x)), f"forward 2^{log_n}"
    i = ctx.ntt(x, inverse=True)
    assert np.array_equal(i, oc.ntt(x, inverse=True)), f"inverse 2^{log_n}"
    assert np.array_equal(ctx.ntt(f, inverse=True), x)


@pytest.mark.parametrize("n", [8, 64, 1024])
def test_ntt_golden(ctx, golden, n):
    g = golden["ntt"][str(n)]
    x = oc.splitmix(n, g["seed"])
    assert ctx.ntt(x).tobytes().hex() == g["forward"]
    assert ctx.ntt(x, inverse=True).tobytes().hex() == g["inverse"]


def test_ntt_4096_u64_inputs_golden(ctx, golden):
    g = golden["ntt"]["4096_u64"]
    x = oc.splitmix(4096, g["seed"])
    x[:, 1] = 0
    assert hashlib.sha256(ctx.ntt(x).tobytes()).hexdigest() == g["forward_sha256"]


def test_ntt_2_22_config(ctx):
    """The headline size: single 2^22-point forward + inverse, bit-exact."""
    n = 1 << 22
    x = oc.splitmix(n, 22)
    f = ctx.ntt(x)
    assert np.array_equal(f, oc.ntt(x))
    assert np.array_equal(ctx.ntt(f, inverse=True), x)


def test_ntt_2_24_full_size(ctx):
    n = 1 << 24
    x = oc.splitmix(n, 24)
    f = ctx.ntt(x)
    ref = oc.ntt(x)
    assert np.array_equal(f, ref)
    del ref
    assert np.array_equal(ctx.ntt(f, inverse=True), x)


def test_ntt_batch(ctx):
    # 2^10 and 2^11: several transforms share a 4096-element tile, the last tile reaching past the batch
    # 2^5 .. 2^8: one pass, 128 .. 16 transforms to a tile; 2^9: a radix-16 pass without exchange, then radix 32
    for log_n, cols in ((5, 7), (13, 5), (16, 3), (10, 1), (10, 7), (10, 8), (11, 3), (11, 130),
                        (5, 1), (5, 128), (5, 129), (6, 65), (7, 1), (7, 33), (8, 16), (8, 17), (8, 1000),
                        (9, 1), (9, 8), (9, 9), (9, 300), (4, 5), (3, 300),
                        # up to 16 points: one transform per thread, the last workgroup partly filled
                        (0, 5), (1, 300), (2, 257), (4, 256), (4, 1000)):
        n = 1 << log_n
        x = oc.splitmix(n * cols, 77 + log_n).reshape(cols, n, 2)
        x[0] = 0                                               # a zero column, and one of p - 1 throughout
        x[-1] = oc.fe(so.P - 1)
        out = ctx.ntt_batch(x)
        for c in range(cols):
            assert np.array_equal(out[c], oc.ntt(x[c]))
        back = ctx.ntt_batch(out, inverse=True)
        assert np.array_equal(back, x)


@pytest.mark.parametrize("log_n,cols", [(2, 300), (4, 70), (5, 40), (8, 19), (9, 11), (10, 5)])
def test_ntt_batch_strided_columns_in_place(ctx, log_n, cols):
    """Short transforms sharing a tile, columns further apart than their length, transformed in place on the
    device: the gaps between the columns must come back untouched."""
    import torch
    n, stride = 1 << log_n, (1 << log_n) + 3
    x = oc.splitmix(stride * cols, 500 + log_n).reshape(cols, stride, 2)
    d = torch.from_numpy(x.view(np.int64).copy()).cuda()
    ctx.ntt_dev(d, n, False, n_cols=cols, col_stride=stride)
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.uint64).reshape(cols, stride, 2)
    for c in range(cols):
        assert np.array_equal(got[c, :n], oc.ntt(np.ascontiguousarray(x[c, :n])))
        assert np.array_equal(got[c, n:], x[c, n:])
    ctx.ntt_dev(d, n, True, n_cols=cols, col_stride=stride)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(cols, stride, 2), x)


def test_ntt_errors(ctx):
    from stark_amd import StarkError
    with pytest.raises(StarkError, match="power of 2"):
        ctx.ntt(oc.splitmix(12, 1))
    with pytest.raises(StarkError) as e:
        ctx.ntt_dev(0, 1 << 31)
    assert e.value.code in (2, 7)


def test_ntt_reference_unit_cases(ctx):
    """ntt.rs:143-206 through the device: evaluate/interpolate round trips and
    (1+x)(1+2x) via pointwise product."""
    x = oc.from_ints([1, 2, 3, 4])
    assert np.array_equal(ctx.ntt(ctx.ntt(x), inverse=True), x)
    a = oc.from_ints([1, 1, 0, 0, 0, 0, 0, 0])
    b = oc.from_ints([1, 2, 0, 0, 0, 0, 0, 0])
    prod = ctx.ntt(ctx.vec_mul(ctx.ntt(a), ctx.ntt(b)), inverse=True)
    assert oc.to_ints(prod)[:3] == [1, 3, 2] and not prod[3:].any()


def test_vec_mul(ctx):
    a, b = _edge_vector(5000, 3), _edge_vector(5000, 4)
    ref = np.empty_like(a)
    oc.lib.so_fe_mul_vec(oc.p64(a), oc.p64(b), oc.p64(ref), 5000)
    assert np.array_equal(ctx.vec_mul(a, b), ref)


# ---- coset LDE (univariate_poly.rs:44-54 on offset*omega^i) ---------------------
@pytest.mark.parametrize("n_coeffs,n_out", [(1, 8), (16, 64), (100, 512), (1024, 4096), (2048, 8192),
                                            (5000, 1 << 15), (1 << 14, 1 << 16), (1 << 18, 1 << 20), (300, 1 << 13)])
def test_coset_lde_vs_oracle(ctx, n_coeffs, n_out):
    c = _edge_vector(n_coeffs, n_coeffs + n_out)
    for off in (G, 1, so.fe_pow(G, 12345)):
        out = ctx.coset_lde_batch(c, n_out, off)[0]
        assert np.array_equal(out, oc.coset_lde(c, n_out, off)), (n_coeffs, n_out, off)


def test_coset_lde_matches_horner_and_golden(ctx, golden):
    g = golden["coset_lde"]
    c = oc.splitmix(g["n_coeffs"], g["seed"])
    out = ctx.coset_lde_batch(c, g["n_out"], int(g["offset"]))[0]
    assert out.tobytes().hex() == g["out"]
    h = np.empty_like(out)
    oc.lib.so_coset_lde_horner(oc.p64(c), len(c), g["n_out"], oc.p64(oc.fe(G)), oc.p64(h))
    assert np.array_equal(out, h)


def test_coset_lde_batch_columns(ctx):
    cols, n_coeffs, n_out = 6, 1 << 12, 1 << 14
    c = oc.splitmix(cols * n_coeffs, 5).reshape(cols, n_coeffs, 2)
    out = ctx.coset_lde_batch(c, n_out, G)
    for k in range(cols):
        assert np.array_equal(out[k], oc.coset_lde(c[k], n_out, G))


@pytest.mark.parametrize("n_coeffs,n_out,cols", [(300, 1024, 5), (1024, 1024, 9), (1, 2048, 3), (2048, 2048, 2), (700, 2048, 131)])
def test_coset_lde_batch_short_columns(ctx, n_coeffs, n_out, cols):
    """Short columns (the local steps of a split transform, FRI-sized batches): transforms of 1024 and
    2048 points share tiles."""
    c = _edge_vector(cols * n_coeffs, n_out + cols).reshape(cols, n_coeffs, 2)
    out = ctx.coset_lde_batch(c, n_out, G)
    for k in range(cols):
        assert np.array_equal(out[k], oc.coset_lde(np.ascontiguousarray(c[k]), n_out, G)), f"column {k}"
    out1 = ctx.coset_lde_batch(c, n_out, 1)
    assert np.array_equal(out1[cols - 1], oc.coset_lde(np.ascontiguousarray(c[cols - 1]), n_out, 1))


def test_coset_lde_full_size_column(ctx):
    """One column of the batched-LDE workload: 2^20 coefficients -> 2^22 points."""
    c = oc.splitmix(1 << 20, 20)
    out = ctx.coset_lde_batch(c, 1 << 22, G)[0]
    assert np.array_equal(out, oc.coset_lde(c, 1 << 22, G))


def test_batched_lde_commit_full_size_64_columns(ctx):
    """The batched workload at its real size (the per-register loop of stark.rs:293-306 over 64
    columns): 64 x 2^20 coefficients -> 64 x 2^22 evaluations on generator * omega^i and one Merkle
    root per column, in one coset_lde_batch_dev and one merkle_roots_fe_batch_dev.  At this size the
    transform runs in scratch-bounded groups of 16 columns and the commitments in groups of trees
    that fit 4 GiB.  Every one of the 64 roots against the single-column commitment of that column (a different
    launch path: one tree per call), and eight columns and their roots against the CPU restatement."""
    import torch
    cols, n_coeffs, n_out = 64, 1 << 20, 1 << 22
    host = oc.splitmix(cols * n_coeffs, 333).reshape(cols, n_coeffs, 2)
    coeffs = torch.from_numpy(host.view(np.int64)).cuda()
    out = torch.empty((cols, n_out, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.coset_lde_batch_dev(coeffs, n_coeffs, n_coeffs, cols, n_out, G, out, n_out)
    roots = ctx.merkle_roots_fe_batch_dev(out, n_out, cols, n_out)       # returns after a synchronisation
    assert len(roots) == cols and len(set(roots)) == cols
    for c in range(cols):
        assert roots[c] == ctx.merkle_commit_fe_dev(out[c], n_out), f"root {c} differs from the single-column commitment"
    with oc.threads(min(16, os.cpu_count() or 1)):
        for c in (0, 1, 15, 16, 31, 32, 47, cols - 1):                   # both ends of every group of 16 columns
            want = oc.coset_lde(np.ascontiguousarray(host[c]), n_out, G)
            got = out[c].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), f"column {c}"
            assert roots[c] == oc.merkle_commit_fe(want), f"root {c}"
    del coeffs, out
    torch.cuda.empty_cache()


# ---- Merkle (merkle.rs) ----------------------------------------------------------
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 10, 11, 12, 13, 16])
def test_merkle_commit_fe_vs_oracle(ctx, log_n):
    n = 1 << log_n
    x = _edge_vector(n, 500 + log_n)
    root, tree = ctx.merkle_commit_fe(x, keep_tree=True)
    ref_tree = oc.merkle_tree_fe(x)
    assert root == ref_tree[-1].tobytes() == oc.merkle_commit_fe(x)
    assert np.array_equal(tree.download(), ref_tree)
    if n > 1:
        for idx in {0, 1, n // 2, n - 1}:
            assert tree.open(idx) == oc.merkle_open(ref_tree, idx)
    tree.free()
    assert ctx.merkle_commit_fe(x) == root


def test_merkle_leaf_encodings(ctx, golden):
    """All five bincode lengths (0,1,2,3,4 digits) hash like the restatement."""
    vals = [int(v) for v, _ in golden["leaf_bytes"]]
    vals += [0] * (16 - len(vals))
    x = oc.from_ints(vals)
    _, tree = ctx.merkle_commit_fe(x, keep_tree=True)
    lv = tree.download()
    for i, v in enumerate(vals):
        assert lv[i].tobytes() == hashlib.blake2b(so.bincode_field_element(v), digest_size=32).digest()


def test_merkle_golden(ctx, golden):
    for n, g in golden["merkle_fe"].items():
        elems = np.frombuffer(bytes.fromhex(g["elems"]), dtype=np.uint64).reshape(-1, 2).copy()
        root, tree = ctx.merkle_commit_fe(elems, keep_tree=True)
        assert root.hex() == g["root"]
        assert [p.hex() for p in tree.open(1)] == g["open_1"]
    g = golden["merkle_bytes"]
    leaves = [bytes.fromhex(l) for l in g["leaves"]]
    root, tree = ctx.merkle_commit_bytes(leaves, keep_tree=True)
    assert root.hex() == g["root"]
    assert [p.hex() for p in tree.open(5)] == g["open_5"]


def test_merkle_reference_unit_case(ctx):
    """merkle.rs:157-206 shape: 64 leaves of 1..99 bytes; open + verify + tamper."""
    n = 64
    leaves = [hashlib.shake_256(bytes([i])).digest(1 + (i * 7) % 99) for i in range(n)]
    root, tree = ctx.merkle_commit_bytes(leaves, keep_tree=True)
    assert root == so.merkle_commit(leaves)
    for i in range(n):
        path = tree.open(i)
        assert so.merkle_verify(root, i, path, leaves[i])
        assert not so.merkle_verify(root, (i + 1) % n, path, leaves[i])
    # long leaves (multi-block BLAKE2b) and empty leaves
    leaves = [b"", b"a" * 128, b"b" * 129, b"c" * 1000]
    assert ctx.merkle_commit_bytes(leaves) == so.merkle_commit(leaves)


def test_merkle_errors(ctx):
    from stark_amd import StarkError
    with pytest.raises(StarkError, match="power of two"):
        ctx.merkle_commit_fe(oc.splitmix(3, 1))
    _, tree = ctx.merkle_commit_fe(oc.splitmix(8, 1), keep_tree=True)
    with pytest.raises(StarkError, match="invalid index"):
        tree.open(8)


def test_merkle_full_size_root(ctx):
    n = 1 << 22
    x = oc.splitmix(n, 9)
    assert ctx.merkle_commit_fe(x) == oc.merkle_commit_fe(x)


@pytest.mark.parametrize("log_n,cols,pad", [(0, 3, 0), (1, 5, 1), (4, 7, 0), (9, 3, 5), (10, 64, 0), (13, 9, 3),
                                            (17, 2, 0), (18, 3, 0)])
def test_merkle_roots_batch_vs_oracle(ctx, log_n, cols, pad):
    """One commit per column of a strided batch (the per-register loop of stark.rs:293-306) equals
    the single-column commit of every column."""
    import torch
    n = 1 << log_n
    stride = n + pad
    host = _edge_vector(cols * stride, 900 + log_n).reshape(cols, stride, 2)
    dev = torch.from_numpy(host.view(np.int64)).cuda()
    roots = ctx.merkle_roots_fe_batch_dev(dev, n, cols, stride)
    assert len(roots) == cols
    for c in range(cols):
        assert roots[c] == oc.merkle_commit_fe(np.ascontiguousarray(host[c, :n]))


def test_merkle_roots_batch_errors(ctx):
    import torch
    from stark_amd import StarkError
    dev = torch.zeros((4, 8, 2), dtype=torch.int64).cuda()
    with pytest.raises(StarkError, match="power of two"):
        ctx.merkle_roots_fe_batch_dev(dev, 6, 2, 8)
    with pytest.raises(StarkError, match="stride"):
        ctx.merkle_roots_fe_batch_dev(dev, 8, 2, 4)
    assert ctx.merkle_roots_fe_batch_dev(dev, 8, 0, 8) == []


# ---- FRI (fri.rs) -------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [1, 2, 8, 9, 13, 18])
def test_fri_fold_vs_oracle(ctx, log_n):
    n = 1 << log_n
    cw = _edge_vector(n, 40 + log_n)
    alpha = so.fe_sample(hashlib.sha256(bytes([log_n])).digest())
    omega = so.primitive_nth_root(n)
    for off in (G, 1):
        assert np.array_equal(ctx.fri_fold(cw, alpha, off, omega), oc.fri_fold(cw, alpha, off, omega))
    # a non-standard root of the right order (omega^3 is also primitive)
    if n >= 8:
        w3 = so.fe_pow(omega, 3)
        assert np.array_equal(ctx.fri_fold(cw, alpha, G, w3), oc.fri_fold(cw, alpha, G, w3))


@pytest.mark.parametrize("log_n", [3, 9, 12])
def test_fri_fold_structured_inputs_and_challenges(ctx, log_n):
    """Folds whose results sit at the ends of the field (0, 1, p - 1) or whose difference a - b vanishes, with
    challenges 0, 1, p - 1 and the one that makes a weight zero: the exact-reduction path behind the lazy sum."""
    n = 1 << log_n
    half = n // 2
    omega = so.primitive_nth_root(n)
    rnd = oc.splitmix(n, 4242 + log_n)
    top = oc.fe(so.P - 1)
    same = rnd.copy()
    same[half:] = same[:half]                                  # a = b everywhere
    ends = np.zeros_like(rnd)
    ends[::2] = top                                            # 0 and p - 1 alternating
    upper = np.zeros_like(rnd)
    upper[half:] = top                                         # a = 0, b = p - 1
    ones = np.zeros_like(rnd)
    ones[:, 0] = 1
    all_top = np.zeros_like(rnd)
    all_top[:] = top
    # alpha = offset makes the weight of b vanish at i = 0 (1 - alpha / x_0 = 0) and doubles nothing else
    for alpha in (0, 1, so.P - 1, G, (so.P + 1) // 2):
        for cw in (np.zeros_like(rnd), all_top, ones, same, ends, upper, rnd):
            got = ctx.fri_fold(np.ascontiguousarray(cw), alpha, G, omega)
            assert np.array_equal(got, oc.fri_fold(np.ascontiguousarray(cw), alpha, G, omega)), (alpha, log_n)


def test_fri_fold_literal_reference_form(ctx):
    n = 64
    cw = oc.splitmix(n, 2)
    alpha = 123456789
    omega = so.primitive_nth_root(n)
    lit = so.fri_fold(oc.to_ints(cw), alpha, G, omega)
    assert oc.to_ints(ctx.fri_fold(cw, alpha, G, omega)) == lit


def _commit_with_python_transcript(ctx, cw, offset, omega, expansion, ncol):
    ps = so.ProofStream()
    alphas = []

    def transcript(rnd, last, root):
        ps.push(root.hex())                                   # fri.rs:120
        if last:
            return None
        a = so.fe_sample(ps.prover_fiat_shamir(32))           # fri.rs:126-127
        alphas.append(a)
        return a

    layers = ctx.fri_commit(cw, offset, omega, expansion, ncol, transcript)
    last = layers.download(len(layers) - 1)
    ps.push(so.json_field_vec(oc.to_ints(last)))              # fri.rs:146
    return layers, ps, alphas


@pytest.mark.parametrize("log_n,expansion,ncol", [(8, 4, 17), (6, 4, 2), (10, 8, 8), (12, 4, 17), (16, 4, 17), (20, 4, 17)])
def test_fri_commit_vs_oracle(ctx, log_n, expansion, ncol):
    n = 1 << log_n
    coeffs = oc.splitmix(n // expansion, 900 + log_n)
    cw = oc.coset_lde(coeffs, n, G)
    omega = so.primitive_nth_root(n)
    layers, ps, alphas = _commit_with_python_transcript(ctx, cw, G, omega, expansion, ncol)
    ref_layers, ref_roots, ref_alphas, ref_stream = oc.fri_commit(cw, G, omega, expansion, ncol)
    assert len(layers) == len(ref_layers) == ctx.fri_num_rounds(n, expansion, ncol)
    assert [layers.root(i) for i in range(len(layers))] == ref_roots
    assert alphas == ref_alphas
    for i in range(len(layers)):
        assert np.array_equal(layers.download(i), ref_layers[i]), f"layer {i}"
    assert ps.objects == ref_stream.objects
    # openings from the retained trees (fri.rs:187-201) and point reads (fri.rs:166-179)
    for i in (0, len(layers) - 1):
        m = layers.layer_len(i)
        ref_tree = oc.merkle_tree_fe(ref_layers[i])
        for idx in (0, m // 2 + 1, m - 1):
            assert layers.open(i, idx) == oc.merkle_open(ref_tree, idx)
        idxs = [0, m - 1, m // 3]
        assert np.array_equal(layers.read(i, idxs), ref_layers[i][idxs])


def test_fri_commit_reference_test_params_golden(ctx, golden):
    g = golden["fri_commit_ref_test"]
    cw = np.frombuffer(bytes.fromhex(g["codeword"]), dtype=np.uint64).reshape(-1, 2).copy()
    omega = so.primitive_nth_root(g["n0"])
    layers, ps, _ = _commit_with_python_transcript(ctx, cw, G, omega, g["expansion"], g["ncol"])
    assert ps.objects == g["objects"]
    assert layers.download(len(layers) - 1).tobytes().hex() == g["last_codeword"]


def test_fri_commit_errors(ctx):
    from stark_amd import StarkError
    cw = oc.splitmix(64, 1)
    with pytest.raises(StarkError, match="less than 1 round"):
        ctx.fri_commit(cw[:4].copy(), G, so.primitive_nth_root(4), 4, 1, lambda *a: 1)
    with pytest.raises(StarkError, match="right order"):
        ctx.fri_commit(cw, G, so.primitive_nth_root(128), 4, 2, lambda *a: 1)

    def boom(*a):
        raise KeyError("transcript")
    with pytest.raises(KeyError):
        ctx.fri_commit(cw, G, so.primitive_nth_root(64), 4, 2, boom)


def test_fri_commit_2_24_full_size(ctx):
    """The FRI headline workload: 2^24 codeword, blowup 4, 17 colinearity tests: 18
    roots, 17 folds.  Checked against the C restatement root by root."""
    n = 1 << 24
    coeffs = oc.splitmix(n // 4, 2424)
    cw = oc.coset_lde(coeffs, n, G)
    omega = so.primitive_nth_root(n)
    layers, ps, alphas = _commit_with_python_transcript(ctx, cw, G, omega, 4, 17)
    assert len(layers) == 18
    ref_layers, ref_roots, ref_alphas, ref_stream = oc.fri_commit(cw, G, omega, 4, 17)
    assert [layers.root(i) for i in range(18)] == ref_roots
    assert alphas == ref_alphas
    assert ps.objects == ref_stream.objects
    assert np.array_equal(layers.download(17), ref_layers[17])
    assert np.array_equal(layers.download(5), ref_layers[5])


# ---- a Merkle tree / a FRI layer cut across devices ------------------------------
@pytest.mark.parametrize("chunk_len,n_chunks", [(1, 8), (2, 1), (64, 16), (1 << 10, 4), (1 << 14, 32), (1 << 18, 2)])
def test_subtree_roots_and_root_of_digests(ctx, chunk_len, n_chunks):
    """Chunk roots equal Merkle::commit of each chunk; finishing the tree from
    them equals Merkle::commit of the whole vector (merkle.rs:38-51)."""
    import torch
    n = chunk_len * n_chunks
    x = oc.splitmix(n, 700 + n_chunks)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    roots = torch.empty((n_chunks, 32), dtype=torch.uint8, device="cuda")
    ctx.merkle_subtree_roots_dev(d, chunk_len, n_chunks, roots)
    ctx.synchronize()
    got = roots.cpu().numpy()
    picks = sorted({0, n_chunks // 2, n_chunks - 1})
    for c in picks:
        assert got[c].tobytes() == oc.merkle_commit_fe(x[c * chunk_len:(c + 1) * chunk_len])
    assert ctx.merkle_root_of_digests_dev(roots, n_chunks) == oc.merkle_commit_fe(x)


def test_root_of_digests_single_and_errors(ctx):
    import torch
    from stark_amd import StarkError
    d = torch.arange(32, dtype=torch.uint8, device="cuda").view(1, 32)
    assert ctx.merkle_root_of_digests_dev(d, 1) == bytes(range(32))
    with pytest.raises(StarkError, match="power of two"):
        ctx.merkle_root_of_digests_dev(d, 3)


@pytest.mark.parametrize("log_n,world,period", [(4, 2, 4), (10, 2, 64), (12, 4, 256), (16, 8, 1 << 10), (20, 4, 1 << 17)])
def test_fri_fold_shard_vs_oracle(ctx, log_n, world, period):
    """Every rank's share of the folded layer, and its chunk roots, equal the
    single-vector fold (fri.rs:133-138) re-dealt the same way."""
    import torch
    from stark_amd.distributed import fri_shard_of
    n = 1 << log_n
    x = oc.splitmix(n, 800 + log_n)
    alpha = oc.to_int(oc.splitmix(1, 9)[0])
    omega, offset = so.primitive_nth_root(n), so.GENERATOR
    ref = oc.fri_fold(x, alpha, offset, omega)
    c = period // world
    full = torch.from_numpy(x.view(np.int64))
    ref_t = torch.from_numpy(ref.view(np.int64))
    for rank in sorted({0, world // 2, world - 1}):
        local = fri_shard_of(full, rank, world, period).cuda()
        out = torch.empty((local.shape[0] // 2, 2), dtype=torch.int64, device="cuda")
        n_out_chunks = out.shape[0] // c
        roots = torch.empty((n_out_chunks, 32), dtype=torch.uint8, device="cuda")
        ctx.fri_fold_shard_dev(local, local.shape[0], alpha, offset, omega, c, period, rank * c, out, roots)
        ctx.synchronize()
        want = fri_shard_of(ref_t, rank, world, period).numpy().view(np.uint64)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
        r = roots.cpu().numpy()
        for q in sorted({0, n_out_chunks - 1}):
            assert r[q].tobytes() == oc.merkle_commit_fe(np.ascontiguousarray(want[q * c:(q + 1) * c]))
        out2 = torch.empty_like(out)
        ctx.fri_fold_shard_dev(local, local.shape[0], alpha, offset, omega, c, period, rank * c, out2)   # no roots
        ctx.synchronize()
        assert torch.equal(out, out2)


def test_fri_fold_shard_errors(ctx):
    import torch
    from stark_amd import StarkError
    d = torch.zeros((64, 2), dtype=torch.int64, device="cuda")
    o = torch.zeros((32, 2), dtype=torch.int64, device="cuda")
    w = so.primitive_nth_root(128)
    with pytest.raises(StarkError, match="powers of two"):
        ctx.fri_fold_shard_dev(d, 64, 1, 1, w, 3, 8, 0, o)
    with pytest.raises(StarkError, match="inside one period"):
        ctx.fri_fold_shard_dev(d, 64, 1, 1, w, 4, 8, 6, o)
    with pytest.raises(StarkError, match="inside one period"):
        ctx.fri_fold_shard_dev(d, 64, 1, 1, w, 16, 8, 0, o)
    with pytest.raises(StarkError, match="even number of chunks"):
        ctx.fri_fold_shard_dev(d, 48, 1, 1, w, 16, 32, 0, o)
    with pytest.raises(StarkError, match="not reduced"):
        ctx.fri_fold_shard_dev(d, 64, so.P, 1, w, 4, 8, 0, o)


def test_library_before_torch_shares_one_hip_runtime():
    """Creating a context before torch touches the GPU must leave torch able to
    initialise it (one HIP runtime per process, whichever is loaded first)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from stark_amd import Context; c = Context(0); import torch; "
            "t = torch.arange(4, device='cuda'); assert int(t.sum()) == 6; c.close(); print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_one_context_from_several_threads(ctx):
    """Calls on one context from several threads are serialised inside the
    library (the reference's tests run on cargo's thread pool): every thread
    gets the results a lone caller gets."""
    import threading
    jobs = [(12, 31), (14, 32), (10, 33), (16, 34), (13, 35), (11, 36)]
    want = {}
    for log_n, seed in jobs:
        x = oc.splitmix(1 << log_n, seed)
        want[seed] = (x, oc.ntt(x), oc.merkle_commit_fe(x))
    errors = []

    def work(log_n, seed):
        try:
            x, f, root = want[seed]
            for _ in range(5):
                assert np.array_equal(ctx.ntt(x), f)
                assert ctx.merkle_commit_fe(x) == root
        except BaseException as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors


@pytest.mark.parametrize("chunk_len,n_chunks", [(1, 4), (8, 2), (256, 16), (1 << 12, 8)])
def test_forest_and_top_tree_give_whole_paths(ctx, chunk_len, n_chunks):
    """Siblings inside a chunk's subtree (kept forest) followed by the siblings
    above its root (kept top tree) are Merkle::open's path in the whole tree."""
    import torch
    from stark_amd import StarkError
    n = chunk_len * n_chunks
    x = oc.splitmix(n, 900 + n_chunks)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    roots = torch.empty((n_chunks, 32), dtype=torch.uint8, device="cuda")
    forest = ctx.merkle_subtree_roots_dev(d, chunk_len, n_chunks, roots, keep_forest=True)
    root, top = ctx.merkle_root_of_digests_dev(roots, n_chunks, keep_tree=True)
    tree = oc.merkle_tree_fe(x)
    assert root == oc.merkle_commit_fe(x)
    for i in sorted({0, 1 % n, n // 2, n - 1, (n // 3) | 1 if n > 2 else 0}):
        assert forest.open(i) + top.open(i // chunk_len) == oc.merkle_open(tree, i)
    if chunk_len > 1 and n_chunks > 1:
        with pytest.raises(StarkError, match="forest"):
            forest.download()
    forest.free()
    top.free()


def test_fold_shard_keeps_forest(ctx):
    import torch
    from stark_amd.distributed import fri_shard_of
    n, world, period, rank = 1 << 12, 4, 256, 2
    c = period // world
    x = oc.splitmix(n, 950)
    alpha = oc.to_int(oc.splitmix(1, 10)[0])
    omega, offset = so.primitive_nth_root(n), so.GENERATOR
    ref = oc.fri_fold(x, alpha, offset, omega)
    tree = oc.merkle_tree_fe(ref)
    local = fri_shard_of(torch.from_numpy(x.view(np.int64)), rank, world, period).cuda()
    out = torch.empty((local.shape[0] // 2, 2), dtype=torch.int64, device="cuda")
    roots = torch.empty((out.shape[0] // c, 32), dtype=torch.uint8, device="cuda")
    forest = ctx.fri_fold_shard_dev(local, local.shape[0], alpha, offset, omega, c, period, rank * c, out, roots,
                                    keep_forest=True)
    ctx.synchronize()
    sub = c.bit_length() - 1
    for l in (0, 5, out.shape[0] - 1):
        pos = (l // c) * period + rank * c + l % c
        assert forest.open(l) == oc.merkle_open(tree, pos)[:sub]
    forest.free()


def test_large_host_transfers_ragged_sizes(ctx):
    """Host-pointer calls from 32 MiB up go through several staging lanes in 4 MiB
    chunks: sizes that are not multiples of a chunk or of the lane count, in and
    out, still give the oracle's results."""
    cols, n_coeffs, n_out = 5, 450001, 1 << 19
    coeffs = oc.splitmix(cols * n_coeffs, 4242).reshape(cols, n_coeffs, 2)
    got = ctx.coset_lde_batch(coeffs, n_out, G)
    for c in range(cols):
        assert np.array_equal(got[c], oc.coset_lde(np.ascontiguousarray(coeffs[c]), n_out, G)), f"column {c}"
    x = oc.splitmix(5 << 19, 4243).reshape(5, 1 << 19, 2)
    f = ctx.ntt_batch(x)
    for c in range(5):
        assert np.array_equal(f[c], oc.ntt(np.ascontiguousarray(x[c])))
    y = oc.splitmix(1 << 21, 4244)
    assert ctx.merkle_commit_fe(y) == oc.merkle_commit_fe(y)


def test_reference_bench_case_2_12_u64_values(ctx):
    """The reference's own bench input: 2^12 points with u64-sized values
    (polynomial_benchmarks.rs:16-21,158), forward then inverse (ntt.rs:25-40)."""
    n = 1 << 12
    rng = np.random.default_rng(12)
    x = np.zeros((n, 2), dtype=np.uint64)
    x[:, 0] = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64)
    f = ctx.ntt(x)
    assert np.array_equal(f, oc.ntt(x))
    assert np.array_equal(ctx.ntt(f, inverse=True), x)


def test_ntt_2_25_four_pass_plan(ctx):
    """Above 2^24 the transform takes four global passes: bit-exact against the oracle."""
    n = 1 << 25
    x = oc.splitmix(n, 25)
    f = ctx.ntt(x)
    assert np.array_equal(f, oc.ntt(x))
    assert np.array_equal(ctx.ntt(f, inverse=True), x)


def _device_field_vector(n, seed):
    """Canonical field elements generated on the device (hi < 2^62 < p >> 64)."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    d = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    d[:, 0] = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), generator=g, device="cuda", dtype=torch.int64)
    d[:, 1] = torch.randint(0, 1 << 62, (n,), generator=g, device="cuda", dtype=torch.int64)
    return d


def _sum_mod_p(d):
    """Sum of the field elements of d (n, 2) modulo p, exact: 32-bit pieces summed in int64."""
    total = 0
    for col, shift in ((0, 0), (1, 64)):
        v = d[:, col]
        low = int((v & 0xFFFFFFFF).sum().item())
        high = int(((v >> 32) & 0xFFFFFFFF).sum().item())
        total += (low + (high << 32)) << shift
    return total % P


@pytest.mark.parametrize("log_n", [28, 30])
def test_ntt_maximum_sizes_properties(ctx, log_n):
    """Up to the largest size the reference admits (2^30, ntt.rs:15-16), where no
    CPU check finishes: X[0] = sum x_j, X[n/2] = sum (-1)^j x_j, and the inverse
    brings every element back."""
    import torch
    n = 1 << log_n
    d = _device_field_vector(n, log_n)
    want0 = _sum_mod_p(d)
    even, odd = _sum_mod_p(d[0::2]), _sum_mod_p(d[1::2])
    keep = d.clone()
    ctx.ntt_dev(d, n, False)
    ctx.synchronize()
    got = d[[0, n // 2]].cpu().numpy().view(np.uint64)
    assert oc.to_int(got[0]) == want0 == (even + odd) % P
    assert oc.to_int(got[1]) == (even - odd) % P
    ctx.ntt_dev(d, n, True)
    ctx.synchronize()
    assert torch.equal(d, keep)
    del d, keep
    torch.cuda.empty_cache()


def test_fri_commit_2_26_spot_checks(ctx):
    """A commit four times the headline size, checked by properties that need no
    full CPU pass: sampled outputs of every fold against fri.rs:133-138 in Python
    integers, and sampled openings of every layer against its root by
    Merkle::verify (merkle.rs:96-132)."""
    import torch
    log_n = 26
    n = 1 << log_n
    d = _device_field_vector(n, 2600)
    omega, offset = so.primitive_nth_root(n), G
    stream, alphas = so.ProofStream(), []

    def transcript(rnd, last, root):
        stream.push(root.hex())
        if last:
            return None
        alphas.append(so.fe_sample(stream.prover_fiat_shamir(32)))
        return alphas[-1]

    layers = ctx.fri_commit(d, offset, omega, 4, 17, transcript, n=n)
    rounds = len(layers)
    assert rounds == ctx.fri_num_rounds(n, 4, 17) == 20
    rng = np.random.default_rng(26)
    two_inv = so.fe_inverse(2)
    om, of = omega, offset
    for r in range(rounds):
        m = layers.layer_len(r)
        assert m == n >> r
        idx = [0, m - 1] + [int(v) for v in rng.integers(0, m, size=2)]
        vals = [oc.to_int(v) for v in layers.read(r, idx)]
        root = layers.root(r)
        for i, v in zip(idx[:3], vals[:3]):
            assert so.merkle_verify(root, i, layers.open(r, i), so.bincode_field_element(v)), (r, i)
        if r + 1 < rounds:
            half = m // 2
            js = [0, half - 1, int(rng.integers(0, half))]
            a = [oc.to_int(v) for v in layers.read(r, js)]
            b = [oc.to_int(v) for v in layers.read(r, [j + half for j in js])]
            out = [oc.to_int(v) for v in layers.read(r + 1, js)]
            for j, x, y, o in zip(js, a, b, out):
                q = so.fe_div(alphas[r], so.fe_mul(of, so.fe_pow(om, j)))
                want = so.fe_mul(two_inv, so.fe_add(so.fe_mul(so.fe_add(1, q), x), so.fe_mul(so.fe_sub(1, q), y)))
                assert o == want, (r, j)
            om, of = so.fe_mul(om, om), so.fe_mul(of, of)
    layers.free()
    del d
    torch.cuda.empty_cache()


def test_open_many_equals_single_opens(ctx):
    """Batched openings (trees and FRI layers) are the single openings, index by index."""
    from stark_amd import StarkError
    n = 1 << 11
    x = oc.splitmix(n, 1234)
    root, tree = ctx.merkle_commit_fe(x, keep_tree=True)
    idx = [0, 1, n - 1, 777, 777, 1024]
    many = tree.open_many(idx)
    ref_tree = oc.merkle_tree_fe(x)
    for i, p in zip(idx, many):
        assert p == tree.open(i) == oc.merkle_open(ref_tree, i)
    assert tree.open_many([]) == []
    with pytest.raises(StarkError, match="cannot open invalid index"):
        tree.open_many([0, n])
    tree.free()
    layers = ctx.fri_commit(x, G, so.primitive_nth_root(n), 4, 3, lambda r, last, root: None if last else 5 + r)
    for layer in (0, len(layers) - 1):
        m = layers.layer_len(layer)
        idx = [0, m - 1, m // 3]
        assert layers.open_many(layer, idx) == [layers.open(layer, i) for i in idx]
    layers.free()
    one = ctx.merkle_commit_fe(x[:1], keep_tree=True)[1]
    assert one.open_many([0]) == [[]]
    one.free()


def test_fri_layer_query_equals_reads_and_opens(ctx):
    """One round of Fri::query in one round trip gives the values and paths of
    the separate reads and openings (fri.rs:155-209)."""
    from stark_amd import StarkError
    n = 1 << 12
    x = oc.splitmix(n, 4321)
    layers = ctx.fri_commit(x, G, so.primitive_nth_root(n), 4, 3, lambda r, last, root: None if last else 11 + r)
    for layer in (0, 3, len(layers) - 2):
        m = layers.layer_len(layer)
        cs = [0, m // 2 - 1, (m // 5) | 1, 7 % (m // 2)]
        vals, pa, pb, pc = layers.query(layer, cs)
        assert np.array_equal(vals[0], layers.read(layer, cs))
        assert np.array_equal(vals[1], layers.read(layer, [c + m // 2 for c in cs]))
        assert np.array_equal(vals[2], layers.read(layer + 1, cs))
        assert pa == [layers.open(layer, c) for c in cs]
        assert pb == [layers.open(layer, c + m // 2) for c in cs]
        assert pc == [layers.open(layer + 1, c) for c in cs]
    with pytest.raises(StarkError, match="cannot open invalid index"):
        layers.query(0, [n // 2])
    with pytest.raises(StarkError, match="bad layer"):
        layers.query(len(layers) - 1, [0])
    layers.free()


@pytest.mark.parametrize("log_n,ncol", [(12, 3), (8, 17), (16, 5)])
def test_fri_query_all_equals_round_by_round(ctx, log_n, ncol):
    """The whole query phase of Fri::prove in one round trip (fri.rs:229-238) against one query per round
    with the indices reduced as the reference reduces them."""
    from stark_amd import StarkError
    n = 1 << log_n
    x = oc.splitmix(n, 99 + log_n)
    layers = ctx.fri_commit(x, G, so.primitive_nth_root(n), 4, ncol, lambda r, last, root: None if last else 5 + 3 * r)
    top = [0, n // 2 - 1, 12345 % (n // 2), (n // 3) | 1, 1]
    got = layers.query_all(top)
    assert len(got) == len(layers) - 1
    for i, (vals, pa, pb, pc) in enumerate(got):
        cs = [t % (layers.layer_len(i) // 2) for t in top]
        v1, a1, b1, c1 = layers.query(i, cs)
        assert np.array_equal(vals, v1) and pa == a1 and pb == b1 and pc == c1
    with pytest.raises(StarkError, match="cannot open invalid index"):
        layers.query_all([n])
    layers.free()


@pytest.mark.parametrize("log_n", [12, 16, 22])
def test_ntt_structured_inputs(ctx, log_n):
    """Inputs whose transforms are full of 0, p-1 and exact multiples of p before
    reduction (constant vectors, a single spike, alternating signs): the
    reductions of lazy values must stay exact (ntt.rs:25-40)."""
    n = 1 << log_n
    cases = {
        "zeros": np.zeros((n, 2), dtype=np.uint64),
        "all p-1": np.tile(oc.fe(P - 1), (n, 1)),
        "all 1": np.tile(oc.fe(1), (n, 1)),
        "spike p-1 at 0": np.zeros((n, 2), dtype=np.uint64),
        "spike 1 at n/2": np.zeros((n, 2), dtype=np.uint64),
        "alternating 1, p-1": np.tile(np.stack([oc.fe(1), oc.fe(P - 1)]), (n // 2, 1)),
    }
    cases["spike p-1 at 0"][0] = oc.fe(P - 1)
    cases["spike 1 at n/2"][n // 2] = oc.fe(1)
    for name, x in cases.items():
        x = np.ascontiguousarray(x)
        for inverse in (False, True):
            assert np.array_equal(ctx.ntt(x, inverse=inverse), oc.ntt(x, inverse=inverse)), (name, inverse)

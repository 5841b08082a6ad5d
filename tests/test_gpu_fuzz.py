"""Seeded random sweeps of the device entry points against the CPU restatement: shapes nobody wrote a
case for (odd column counts and strides, partly filled columns, every transform length up to 2^15 in
both directions, in place and out of place).  Bit-exact like every other parity test."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_c as oc          # noqa: E402
import stark_oracle as so      # noqa: E402

P, G = so.P, so.GENERATOR
SEED = int(os.environ.get("STARK_FUZZ_SEED", "0"))       # another sweep: STARK_FUZZ_SEED=1 pytest tests/test_gpu_fuzz.py


@pytest.fixture(scope="module")
def ctx():
    from stark_amd import Context
    c = Context(0)
    yield c
    c.close()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()


def _host(t, shape):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint64).reshape(shape)


def test_random_batched_transforms(ctx):
    """stark_ntt_batch_dev: random length 2^0 .. 2^15, column count, stride and direction; the gaps between
    the columns must come back untouched and the inverse must give the input back."""
    rng = random.Random(20261111 + SEED)
    for case in range(140):
        log_n = rng.randrange(0, 16)
        n = 1 << log_n
        cols = rng.choice([1, 2, 3, 5, 8, 17, 33, 100, 257]) if log_n <= 10 else rng.choice([1, 2, 3, 7])
        stride = n + rng.choice([0, 0, 1, 3, 16])
        inverse = bool(rng.getrandbits(1))
        x = oc.splitmix(stride * cols, 9000 + case + 100000 * SEED).reshape(cols, stride, 2)
        if case % 9 == 0:                                  # structured columns: zeros, ones, p - 1
            x[0, :n] = 0
            x[-1, :n] = oc.fe(P - 1)
            if cols > 2:
                x[1, :n] = oc.fe(1)
        d = _dev(x)
        ctx.ntt_dev(d, n, inverse, n_cols=cols, col_stride=stride)
        got = _host(d, (cols, stride, 2))
        for c in range(cols):
            assert np.array_equal(got[c, :n], oc.ntt(np.ascontiguousarray(x[c, :n]), inverse)), (case, log_n, cols, stride, inverse, c)
            assert np.array_equal(got[c, n:], x[c, n:]), (case, log_n, cols, stride)
        ctx.ntt_dev(d, n, not inverse, n_cols=cols, col_stride=stride)
        assert np.array_equal(_host(d, (cols, stride, 2)), x), (case, log_n, cols, stride, inverse)


def test_random_coset_extensions(ctx):
    """stark_coset_lde_batch_dev: random coefficient count (not a power of two, down to one), output
    length, column count, strides and offset, against Polynomial::eval_domain on offset * omega^i."""
    rng = random.Random(77 + SEED)
    for case in range(70):
        log_out = rng.randrange(0, 15)
        n_out = 1 << log_out
        n_coeffs = rng.randrange(1, n_out + 1)
        cols = rng.choice([1, 2, 3, 6, 19]) if log_out <= 11 else rng.choice([1, 2, 3])
        c_stride = n_coeffs + rng.choice([0, 0, 2, 5])
        o_stride = n_out + rng.choice([0, 0, 1, 8])
        offset = rng.choice([1, G, 2, P - 1, rng.randrange(1, P)])
        coeffs = oc.splitmix(c_stride * cols, 400 + case + 100000 * SEED).reshape(cols, c_stride, 2)
        out = oc.splitmix(o_stride * cols, 800 + case).reshape(cols, o_stride, 2)
        d_c, d_o = _dev(coeffs), _dev(out)
        ctx.coset_lde_batch_dev(d_c, n_coeffs, c_stride, cols, n_out, offset, d_o, o_stride)
        got = _host(d_o, (cols, o_stride, 2))
        for c in range(cols):
            ref = oc.coset_lde(np.ascontiguousarray(coeffs[c, :n_coeffs]), n_out, offset)
            assert np.array_equal(got[c, :n_out], ref), (case, n_coeffs, n_out, cols, offset, c)
            assert np.array_equal(got[c, n_out:], out[c, n_out:]), (case, n_coeffs, n_out)
        assert np.array_equal(_host(d_c, (cols, c_stride, 2)), coeffs)       # the coefficients are only read


def test_random_trees_and_folds(ctx):
    """Merkle::commit over field elements and one FRI fold at random lengths; root, every opened path and
    the folded codeword against the restatement."""
    rng = random.Random(5 + SEED)
    for case in range(40):
        log_n = rng.randrange(0, 15)
        n = 1 << log_n
        x = oc.splitmix(n, 1200 + case + 100000 * SEED)
        if case % 5 == 0:
            x[rng.randrange(n)] = oc.fe(rng.choice([0, 1, 255, 256, (1 << 32) - 1, 1 << 32, 1 << 64, (1 << 96) - 1, P - 1]))
        root, tree = ctx.merkle_commit_fe(x, keep_tree=True)
        ref_tree = oc.merkle_tree_fe(x)
        assert root == oc.merkle_commit_fe(x), (case, log_n)
        # (a tree of one leaf has nothing to open: merkle.rs:54-76 recurses without end on it)
        for i in {0, n - 1, rng.randrange(n), rng.randrange(n)} if n > 1 else ():
            assert tree.open(i) == oc.merkle_open(ref_tree, i), (case, log_n, i)
        tree.free()
        if log_n >= 1:
            alpha, omega = rng.randrange(P), pow(G, (1 << 119) >> log_n, P)
            offset = rng.choice([G, 1, rng.randrange(1, P)])
            assert np.array_equal(ctx.fri_fold(x, alpha, offset, omega), oc.fri_fold(x, alpha, offset, omega)), (case, log_n)


def test_repeated_calls_do_not_grow_device_memory():
    """The context's caches are bounded: after a first round of calls has built every table and filled the
    block cache, a few hundred more rounds of the same calls leave the device's free memory where it was."""
    import torch
    from stark_amd import Context
    c = Context(0)
    try:
        n = 1 << 14
        x = oc.splitmix(n, 3)
        omega = pow(G, (1 << 119) >> 14, P)

        def one_round(k):
            f = c.ntt(x)
            c.ntt(f, inverse=True)
            cw = c.coset_lde_batch(x[: n // 4], n, G if k % 2 else 3 + k % 5)[0]     # a few distinct offsets
            root, tree = c.merkle_commit_fe(cw, keep_tree=True)
            tree.open(k % n)
            tree.free()
            layers = c.fri_commit(cw, G, omega, 4, 5, lambda r, last, rt: None if last else 7 + r)
            layers.query_all([1, 2, 3])
            layers.free()

        for k in range(12):
            one_round(k)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for k in range(300):
            one_round(k)
        torch.cuda.synchronize()
        free_after = torch.cuda.mem_get_info()[0]
        assert free_before - free_after < (64 << 20), (free_before, free_after)
    finally:
        c.close()

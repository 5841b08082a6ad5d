"""GPU side of the multi-GPU path on the one GPU a test box has: the four-step
transform with a 1-rank RCCL group (exercises HipEngine, the twiddle kernel
and the exchange code on device tensors) against the C restatement."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_c as oc          # noqa: E402
import stark_oracle as so      # noqa: E402


@pytest.fixture(scope="module")
def group():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    port = 29900 + os.getpid() % 1000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def engine():
    import torch
    from stark_amd import Context
    from stark_amd.distributed import HipEngine
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    yield HipEngine(ctx)
    ctx.close()


def test_fourstep_twiddle_kernel(engine):
    import torch
    n_total, rows, cols, row_start = 1 << 16, 24, 256, 100
    x = oc.splitmix(rows * cols, 4).reshape(rows, cols, 2)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    for inverse in (False, True):
        tt = t.clone()
        engine.twiddle(tt, n_total, row_start, inverse)
        torch.cuda.synchronize()
        got = tt.cpu().numpy().view(np.uint64)
        w = so.primitive_nth_root(n_total)
        if inverse:
            w = so.fe_inverse(w)
        for i in (0, 7, rows - 1):
            for k in (0, 1, 17, cols - 1):
                exp = so.fe_mul(oc.to_int(x[i, k]), so.fe_pow(w, (row_start + i) * k))
                assert oc.to_int(got[i, k]) == exp


@pytest.mark.parametrize("log_n", [14, 16, 20])
def test_fourstep_single_rank_matches_ntt(group, engine, log_n):
    import torch
    from stark_amd.distributed import fourstep_ntt
    n = 1 << log_n
    x = oc.splitmix(n, 60 + log_n)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    y = fourstep_ntt(t, n, engine, inverse=False)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint64), oc.ntt(x))
    z = fourstep_ntt(y.contiguous(), n, engine, inverse=True)
    torch.cuda.synchronize()
    assert np.array_equal(z.cpu().numpy().view(np.uint64), x)


def test_column_sharded_lde_commit_single_rank(group, engine):
    import torch
    from stark_amd.distributed import lde_commit_columns, shard_columns
    n_cols, n_coeffs, n_out = 3, 1 << 10, 1 << 12
    coeffs = oc.splitmix(n_cols * n_coeffs, 8).reshape(n_cols, n_coeffs, 2)
    assert shard_columns(n_cols, 0, 1) == [0, 1, 2]
    cw, roots = lde_commit_columns(engine, torch.from_numpy(coeffs.view(np.int64)).cuda(), n_cols, n_out, so.GENERATOR)
    torch.cuda.synchronize()
    for c in range(n_cols):
        ref = oc.coset_lde(coeffs[c], n_out, so.GENERATOR)
        assert np.array_equal(cw[c].cpu().numpy().view(np.uint64), ref)
        assert roots[c] == oc.merkle_commit_fe(ref)

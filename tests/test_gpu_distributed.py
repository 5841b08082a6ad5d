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


def test_fri_commit_sharded_single_rank(group, engine):
    """One rank, several chunks per period is not possible (chunk = period), but
    the sharded rounds, the exchange code on device tensors and the hand-over to
    the whole-layer commit all run; results equal the single-vector commit."""
    import torch
    from stark_amd.distributed import fri_commit_sharded
    n, expansion, ncol = 1 << 16, 4, 17
    x = oc.splitmix(n, 77)
    omega, offset = so.primitive_nth_root(n), so.GENERATOR
    ref_layers, ref_roots, ref_alphas, _ = oc.fri_commit(x, offset, omega, expansion, ncol)
    stream, alphas = oc.Stream(), []

    def transcript(rnd, last, root):
        stream.push(root.hex())
        if last:
            return None
        alphas.append(so.fe_sample(stream.fiat_shamir(32)))
        return alphas[-1]

    res = fri_commit_sharded(engine, torch.from_numpy(x.view(np.int64)).cuda(), n, offset, omega, expansion, ncol,
                             transcript, period=1 << 10)
    torch.cuda.synchronize()
    assert len(res.shares) == 6
    assert res.roots == ref_roots and alphas == ref_alphas
    for k in range(len(res.shares)):
        assert np.array_equal(res.gather_layer(k).cpu().numpy().view(np.uint64), ref_layers[k])
    for i in range(len(res.tail)):
        assert np.array_equal(res.tail.download(i), ref_layers[len(res.shares) + i])
    res.tail.free()


def _fri_two_rank_worker(rank, world, port, q):
    import sys
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        for p in (os.path.dirname(here), here, os.path.join(os.path.dirname(here), "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle_c
        import stark_oracle
        from stark_amd import Context
        from stark_amd.distributed import HipEngine, fri_commit_sharded, fri_shard_of, fri_shard_period
        torch.cuda.set_device(0)
        ctx = Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        eng = HipEngine(ctx)
        n, expansion, ncol = 1 << 18, 4, 17
        x = oracle_c.splitmix(n, 78)
        omega, offset = stark_oracle.primitive_nth_root(n), stark_oracle.GENERATOR
        ref_layers, ref_roots, ref_alphas, _ = oracle_c.fri_commit(x, offset, omega, expansion, ncol)
        period = fri_shard_period(n, world, expansion, ncol, 1 << 12)
        local = fri_shard_of(torch.from_numpy(x.view(np.int64)), rank, world, period).cuda()
        stream, alphas = oracle_c.Stream(), []

        def transcript(rnd, last, root):
            stream.push(root.hex())
            if last:
                return None
            alphas.append(stark_oracle.fe_sample(stream.fiat_shamir(32)))
            return alphas[-1]

        res = fri_commit_sharded(eng, local, n, offset, omega, expansion, ncol, transcript, period=1 << 12)
        torch.cuda.synchronize()
        ok = res.roots == ref_roots and alphas == ref_alphas and len(res.shares) == 6
        for k in range(len(res.shares)):
            ok = ok and np.array_equal(res.gather_layer(k).cpu().numpy().view(np.uint64), ref_layers[k])
        for i in range(len(res.tail)):
            ok = ok and np.array_equal(res.tail.download(i), ref_layers[len(res.shares) + i])
        res.tail.free()
        ctx.close()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_fri_commit_sharded_two_ranks_one_gpu():
    """Two processes share the one GPU of a test box (each with its own context
    and HIP kernels; the 32-byte exchanges go through gloo): roots, challenges
    and layers equal the single-vector commit."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 30700 + os.getpid() % 1000
    procs = [mpc.Process(target=_fri_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _fri_prove_two_rank_worker(rank, world, port, q):
    import sys
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        for p in (os.path.dirname(here), here, os.path.join(os.path.dirname(here), "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle_c
        import stark_oracle
        from stark_amd import Context, hostlib
        from stark_amd.distributed import HipEngine, fri_prove_sharded, fri_shard_of, fri_shard_period
        torch.cuda.set_device(0)
        ctx = Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        eng = HipEngine(ctx)
        n, expansion, ncol = 1 << 13, 4, 5
        coeffs = oracle_c.splitmix(n // expansion, 654)
        omega, offset = stark_oracle.primitive_nth_root(n), stark_oracle.GENERATOR
        x = oracle_c.coset_lde(coeffs, n, offset)
        ref_stream = stark_oracle.ProofStream()
        fri = stark_oracle.Fri(offset, omega, n, expansion, ncol)
        ref_indices = fri.prove(oracle_c.to_ints(x), ref_stream)
        period = fri_shard_period(n, world, expansion, ncol, 1 << 8)
        local = fri_shard_of(torch.from_numpy(x.view(np.int64)), rank, world, period).cuda()
        stream = hostlib.Stream()                       # the C++ ProofStream of the host mirror
        indices = fri_prove_sharded(eng, local, n, offset, omega, expansion, ncol, stream, period=1 << 8)
        torch.cuda.synchronize()
        same = list(stream.objects) == ref_stream.objects and indices == ref_indices
        accepted = fri.verify(stark_oracle.ProofStream.deserialize(stream.serialize()), [])
        ctx.close()
        q.put((rank, bool(same), bool(accepted)))
    finally:
        dist.destroy_process_group()


def test_fri_prove_sharded_two_ranks_one_gpu():
    """Sharded Fri::prove with the HIP engine on two processes sharing the GPU:
    the proof stream equals the single-vector prover's restatement object for
    object and its verifier accepts."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 30900 + os.getpid() % 1000
    procs = [mpc.Process(target=_fri_prove_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=280) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, True), (1, True, True)]


# ---- the local steps around the all-to-all as HIP kernels ----------------------------------------
@pytest.mark.parametrize("rows,cols,world", [(1, 8, 2), (24, 256, 4), (100, 96, 3), (512, 4096, 8), (33, 70, 1)])
def test_fourstep_pack_and_unpack_kernels(engine, rows, cols, world):
    import torch
    x = oc.splitmix(rows * cols, rows + cols).reshape(rows, cols, 2)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    send = engine.pack(t, world)
    torch.cuda.synchronize()
    cpw = cols // world
    want = x.reshape(rows, world, cpw, 2).transpose(1, 2, 0, 3)                 # [dest][column][row]
    assert np.array_equal(send.cpu().numpy().view(np.uint64), want)
    # with the twiddle of the exchange that follows the first set of transforms
    n_total, row_start = 1 << 22, 37
    for inverse in (False, True):
        send = engine.pack(t, world, (n_total, row_start, inverse))
        torch.cuda.synchronize()
        got = send.cpu().numpy().view(np.uint64)
        w = so.primitive_nth_root(n_total)
        if inverse:
            w = so.fe_inverse(w)
        for r in (0, rows // 2, rows - 1):
            for c in (0, 1, cols // 2, cols - 1):
                exp = so.fe_mul(oc.to_int(x[r, c]), so.fe_pow(w, (row_start + r) * c))
                assert oc.to_int(got[c // cpw, c % cpw, r]) == exp
    # unpack: [source][row][run] -> [row][source][run]
    run = 5
    y = oc.splitmix(world * rows * run, 9).reshape(world, rows, run, 2)
    out = engine.unpack(torch.from_numpy(y.view(np.int64)).cuda(), world, rows, run)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), y.transpose(1, 0, 2, 3).reshape(rows, world * run, 2))


def test_fourstep_under_another_torch_stream(group, engine):
    """The engine follows torch's current stream: tensors produced and consumed on a side stream."""
    import torch
    from stark_amd.distributed import fourstep_ntt
    n = 1 << 16
    x = oc.splitmix(n, 160)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t = torch.from_numpy(x.view(np.int64)).cuda()
        y = fourstep_ntt(t, n, engine, inverse=False)
        z = fourstep_ntt(y.contiguous(), n, engine, inverse=True)
    side.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint64), oc.ntt(x))
    assert np.array_equal(z.cpu().numpy().view(np.uint64), x)
    y2 = fourstep_ntt(torch.from_numpy(x.view(np.int64)).cuda(), n, engine)        # and back on the default stream
    torch.cuda.synchronize()
    assert np.array_equal(y2.cpu().numpy().view(np.uint64), oc.ntt(x))


def _fourstep_two_rank_worker(rank, world, port, q):
    import sys
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        for p in (os.path.dirname(here), here, os.path.join(os.path.dirname(here), "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle_c
        from stark_amd import Context
        from stark_amd.distributed import HipEngine, fourstep_ntt, fourstep_split
        torch.cuda.set_device(0)
        ctx = Context(0)
        eng = HipEngine(ctx)
        ok = True
        for log_n in (16, 20):
            n = 1 << log_n
            x = oracle_c.splitmix(n, 500 + log_n)
            n1, n2 = fourstep_split(n, world)
            share = n // world
            local = torch.from_numpy(x[rank * share:(rank + 1) * share].view(np.int64)).cuda()
            for inverse in (False, True):
                ref = oracle_c.ntt(x, inverse)
                y = fourstep_ntt(local, n, eng, inverse=inverse)
                torch.cuda.synchronize()
                ok = ok and np.array_equal(y.cpu().numpy().view(np.uint64), ref[rank * share:(rank + 1) * share])
                # without the last exchange: rows k1 (this rank's block) x columns k2, X[k1 + n1*k2]
                yt = fourstep_ntt(local, n, eng, inverse=inverse, natural_output=False)
                torch.cuda.synchronize()
                rows = n1 // world
                want = ref.reshape(n2, n1, 2).transpose(1, 0, 2)[rank * rows:(rank + 1) * rows]
                ok = ok and np.array_equal(yt.cpu().numpy().view(np.uint64), want)
        ctx.close()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_fourstep_two_ranks_one_gpu():
    """The four-step transform with the HIP engine on two processes sharing the GPU (gloo carries the
    all-to-all): 2^16 and 2^20 points, forward and inverse, natural and transposed output, against the
    restatement of NTT::forward / inverse (ntt.rs:25-40) on the whole vector."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 31100 + os.getpid() % 1000
    procs = [mpc.Process(target=_fourstep_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=280) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_sharded_prove_over_a_one_rank_rccl_group(group):
    """The prover spread over ranks with RCCL carrying the exchanges on device tensors (all-to-all of the
    extension, all-gather of the subtree roots), as far as one GPU can show it: a group of one rank.  The proof is
    the plain prover's."""
    import hashlib
    import torch
    from stark_amd import hostlib, scaled
    from stark_amd import sharded_stark as S
    from stark_amd.distributed import HipEngine
    assert group.get_backend() == "nccl"
    ctx = scaled.mirror_context()
    eng = HipEngine(ctx)
    job = scaled.ChainJob(99, 1 << 12, 4, 2, 2)
    random = hashlib.shake_256(b"one rank of rccl").digest(job.random_bytes)
    ref, _ = job.prove(random)
    steps = scaled.ChainSteps(job, random, ctx)
    proof = S.stark_prove_sharded(eng, steps, hostlib.Stream(), period=1 << 10)
    steps.free()
    torch.cuda.synchronize()
    assert proof == ref and job.verify(proof)
    job.free()

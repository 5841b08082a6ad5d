"""The prover spread over ranks (stark_amd/sharded_stark.py) with the HIP engine: two processes share the one GPU
of a test box (gloo carries the exchanges).  On every rank the proof must be the single-GPU prover's
(ScaledStark::prove from the same random bytes), byte for byte; the mirror's verifier and the restatement's accept."""
import hashlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [(1 << 8, 256), (1 << 10, 1 << 9), (1 << 14, 1 << 12), (1000, 64)]      # (trace rows, period of the chunked layout)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["STARK_DEVICE"] = "0"                       # both ranks on the box's one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        from stark_amd import hostlib, scaled
        from stark_amd import sharded_stark as S
        from stark_amd.distributed import HipEngine
        torch.cuda.set_device(0)
        ctx = scaled.mirror_context()
        eng = HipEngine(ctx)
        results = []
        for rows, period in CASES:
            job = scaled.ChainJob(2024, rows, 4, 2, 2)
            random = hashlib.shake_256(b"sharded on the gpu %d" % rows).digest(job.random_bytes)
            ref, _ = job.prove(random)
            by_columns, _ = job.prove(random, by_columns=True)
            steps = scaled.ChainSteps(job, random, ctx)
            timings = {}
            proof = S.stark_prove_sharded(eng, steps, hostlib.Stream(), period=period, timings=timings)
            steps.free()
            torch.cuda.synchronize()
            ok = proof == ref and by_columns == ref and job.verify(proof) and timings["total"] > 0
            if rank == 0 and rows <= 1 << 10:               # the restatement's verifier (Python: short traces only)
                import stark_prover_oracle as sp
                import stark_scaled_oracle as ss
                rp = sp.RescuePrime()
                st = ss.ScaledStark(4, 2, 2, rp.m, rows)
                ok = ok and st.verify(proof, ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows),
                                      ss.rescue_chain_boundary(rp, rows, job.output()))
            results.append(bool(ok))
            job.free()
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_prove_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 31300 + os.getpid() % 1000
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, [True] * len(CASES)), (1, [True] * len(CASES))]


def test_sharded_prove_single_rank_is_the_plain_prover():
    """Without a process group the same code path runs on one GPU (one chunk per period) and gives the same bytes."""
    import torch
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from stark_amd import hostlib, scaled
    from stark_amd import sharded_stark as S
    from stark_amd.distributed import HipEngine
    torch.cuda.set_device(0)
    ctx = scaled.mirror_context()
    eng = HipEngine(ctx)
    job = scaled.ChainJob(7, 1 << 12, 4, 2, 2)
    random = hashlib.shake_256(b"one rank").digest(job.random_bytes)
    ref, _ = job.prove(random)
    steps = scaled.ChainSteps(job, random, ctx)
    proof = S.stark_prove_sharded(eng, steps, hostlib.Stream(), period=1 << 10)
    steps.free()
    assert proof == ref and job.verify(proof)
    job.free()

"""world_size-2 and -4 gloo runs of the prover spread over ranks (stark_amd/sharded_stark.py) on CPU: the local
arithmetic is the restatement's, the exchanges, layouts and the assembly of the proof are the product's.  The
proof on every rank must equal the single-process prover's, byte for byte, and the restated verifier accepts it."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from test_distributed_cpu import OracleEngine  # noqa: E402


class Engine(OracleEngine):
    def compose_linear(self, t, s):
        a = t.numpy().view(np.uint64)
        for c in range(a.shape[0]):
            vals = self.oc.to_ints(np.ascontiguousarray(a[c]))
            w, out = 1, []
            for v in vals:
                out.append(v * w % self.so.P)
                w = w * s % self.so.P
            a[c] = self.oc.from_ints(out)


class OracleSteps:
    """The prover's steps off the FRI domain from the restatement (oracle/stark_scaled_oracle.py), as tensors."""

    def __init__(self, stark, trace, air, constants, boundary, rng, oc):
        self.st, self.air, self.constants, self.boundary, self.rng, self.oc = stark, air, constants, boundary, rng, oc
        self.state = stark.step_begin(trace, boundary, rng)
        self.fri_domain_length = stark.fri_domain_length
        self.expansion_factor, self.num_colinearity_tests = stark.expansion_factor, stark.num_colinearity_tests
        self.offset, self.omega = stark.generator, stark.omega

    def _tensor(self, polys):
        n = max(len(p) for p in polys)
        rows = [self.oc.from_ints(list(p) + [0] * (n - len(p))) for p in polys]
        return torch.from_numpy(np.stack(rows).view(np.int64))

    def boundary_quotients(self):
        return self._tensor(self.state["boundary_quotients"])

    def randomizer(self):
        self.st.step_transition(self.state, self.air, self.constants, self.rng)
        return self._tensor([self.state["randomizer_poly"]])

    def combination(self, seed):
        return self._tensor([self.st.step_combine(self.state, self.air, self.boundary, seed)])

    def finish(self):
        pass


def _statement(rows, expansion, ncol):
    import stark_prover_oracle as sp
    import stark_scaled_oracle as ss
    rp = sp.RescuePrime()
    st = ss.ScaledStark(expansion, ncol, 2, rp.m, rows)
    trace = ss.rescue_chain_trace(rp, 2024, rows)
    air = ss.rescue_column_air(rp)
    constants = ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    n_random = 17 * (st.num_randomizers * rp.m + st.max_degree(air) + 1)
    return ss, st, trace, air, constants, boundary, n_random


def _worker(rank, world, port, rows, expansion, ncol, period, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stark_oracle as so
        from stark_amd import sharded_stark as S
        eng = Engine()
        ss, st, trace, air, constants, boundary, n_random = _statement(rows, expansion, ncol)
        seed = b"sharded %d" % rows
        ref = st.prove(trace, air, constants, boundary, ss.stream_rng(seed, n_random)) if rank == 0 else None

        class Stream:                                          # proof_stream.rs:23-58
            def __init__(self):
                self.ps = so.ProofStream()

            def push(self, s):
                self.ps.push(s)

            def fiat_shamir(self, nbytes):
                return self.ps.prover_fiat_shamir(nbytes)

            def serialize(self):
                return self.ps.serialize()

        steps = OracleSteps(st, trace, air, constants, boundary, ss.stream_rng(seed, n_random), eng.oc)
        proof = S.stark_prove_sharded(eng, steps, Stream(), period=period)
        gathered = [None] * world
        dist.all_gather_object(gathered, proof)
        same_everywhere = all(g == proof for g in gathered)
        if rank == 0:
            q.put((proof == ref, same_everywhere, bool(st.verify(proof, air, constants, boundary))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,expansion,ncol,period", [
    (2, 1 << 8, 4, 2, 256),        # FRI domain 4096: 16 periods, 4 sharded FRI rounds
    (4, 1 << 8, 4, 2, 64),
    (2, 1 << 10, 4, 2, None),      # default period: longer than the codeword, so everything is replicated
    (4, 100, 8, 3, 32),            # odd trace length, other parameters
    (2, 1 << 10, 4, 2, 1 << 10),
])
def test_sharded_prove_equals_single_prover(world, rows, expansion, ncol, period):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + world * 17 + rows % 97
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, expansion, ncol, period, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, everywhere, accepted = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same, "the sharded prover's proof differs from the single-process prover's"
    assert everywhere, "ranks ended with different proofs"
    assert accepted, "the verifier rejected the sharded prover's proof"


def test_lde_to_shares_single_process():
    """Without a process group the extension is the plain one, in the chunked layout of one rank."""
    from stark_amd import distributed as D
    from stark_amd import sharded_stark as S
    eng = Engine()
    oc, so = eng.oc, eng.so
    coeffs = oc.splitmix(2 * 48, 11).reshape(2, 48, 2)
    got = S.lde_to_shares(eng, torch.from_numpy(coeffs.view(np.int64)), 256, so.GENERATOR, 64)
    for c in range(2):
        ref = oc.coset_lde(np.ascontiguousarray(coeffs[c]), 256, so.GENERATOR)
        assert np.array_equal(got[c].numpy().view(np.uint64), ref)
    assert S.json_field_element(0) == so.json_field_element(0)
    assert S.json_field_element(so.P - 1) == so.json_field_element(so.P - 1)
    assert D.fri_shard_period(4096, 2, 4, 2, 256) == 256

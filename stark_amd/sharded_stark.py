"""Stark::prove (stark.rs:223-471) for a long trace with the FRI-domain work spread
over the GPUs of a node: one process per GPU, torch.distributed for the exchanges.

Every rank runs the same steps of the prover and the same transcript.  What is
not spread are the steps whose vectors are short next to the FRI domain (trace
interpolation, quotients: every rank computes them alike from the same trace and
random bytes).  What is spread is everything on the FRI domain, whose vectors
are `expansion_factor` times longer:

* low-degree extensions (stark.rs:291-297,356,406): a four-step transform whose
  input is on every rank already, so of the three exchanges of the general
  transform only the middle one remains, and whose output lands in the chunked
  layout the sharded FRI folds in place (distributed.fri_shard_of);
* commitments (stark.rs:302-305,358-360): each rank reduces its chunks to
  subtree roots, the roots are all-gathered, every rank finishes the top levels;
* the nonlinear combination's FRI proof (stark.rs:409): distributed.fri_prove_sharded;
* openings of the boundary-quotient and randomizer codewords (stark.rs:429-464):
  the owner reads the leaf and the siblings inside its subtree, one all-reduce
  per codeword carries them to every rank, the upper siblings are replicated.

Every rank ends with the same proof, byte for byte the single-GPU prover's from
the same trace and random bytes."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .distributed import (_a2a_transpose, _all_gather, _u32_digits, fri_prove_sharded, fri_shard_of, fri_shard_period,
                          json_hash_path, open_in_shares)


def json_field_element(v: int) -> str:
    """serde_json::to_string(&FieldElement) (stark.rs:433,455): BigInt as [sign, u32 digits low first]."""
    return '{"value":[%d,[%s]]}' % (0 if v == 0 else 1, ",".join(map(str, _u32_digits(v))))


def _world_rank(group):
    if dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def lde_to_shares(engine, coeffs: torch.Tensor, n_out: int, offset: int, period: int, group=None) -> torch.Tensor:
    """Polynomial::eval_domain on offset * w^i, i < n_out (univariate_poly.rs:44-54), of `cols` polynomials whose
    coefficients (cols, n_coeffs, 2), lowest first, every rank holds; each rank gets its share of every codeword
    in the chunked layout of fri_shard_of(., rank, world, period): (cols, n_out / world, 2).

    With Q = n_out / period periods, position i = q * period + m and coefficient k = k_hi * Q + k_lo:
      X[q * period + m] = sum_{k_lo} wQ^(q k_lo) * [ w^(m k_lo) * sum_{k_hi} c'_k wM^(m k_hi) ],  c'_k = c_k offset^k.
    A rank takes Q / world of the k_lo (rows of the coefficient matrix, strided in the vector) and transforms
    them over k_hi; ONE all-to-all hands every rank the columns m of its own chunk of every period, multiplied by
    w^(m k_lo) on the way; transforms over k_lo and a local transposition leave X[q * period + rank * C + j] at
    local index q * C + j.  When there are fewer periods than ranks (short codewords) every rank evaluates the
    whole codeword and keeps its share."""
    world, rank = _world_rank(group)
    cols, n_coeffs, _ = coeffs.shape
    q = n_out // period
    c = period // world
    if not dist.is_initialized() or q % world:
        cw = engine.coset_lde(coeffs, n_out, offset)
        return torch.stack([fri_shard_of(cw[k], rank, world, period) for k in range(cols)])
    mc = -(-n_coeffs // q)                                            # k_hi < mc <= period
    scaled = torch.zeros((cols, mc * q, 2), dtype=coeffs.dtype, device=coeffs.device)
    scaled[:, :n_coeffs] = coeffs
    engine.compose_linear(scaled, offset)                             # c_k * offset^k
    mine = scaled.view(cols, mc, world, q // world, 2)[:, :, rank].permute(0, 2, 1, 3).contiguous()
    y = engine.coset_lde(mine.view(cols * (q // world), mc, 2), period, 1).view(cols, q // world, period, 2)
    shares = []
    for k in range(cols):
        z = _a2a_transpose(y[k], group, engine, twiddle=(n_out, rank * (q // world), False))      # (C, Q, 2)
        engine.ntt_batch(z, False)
        zt = engine.transpose(z) if hasattr(engine, "transpose") else z.permute(1, 0, 2).contiguous()
        shares.append(zt.reshape(q * c, 2))
    return torch.stack(shares)


class ShardedCommitment:
    """Merkle::commit (merkle.rs:38-51) of a vector dealt out in chunks: the root, this rank's share and its
    subtrees (`forest`), and the levels above the subtree roots (`top`, on every rank)."""

    def __init__(self, root: bytes, share: torch.Tensor, forest, top, chunk_len: int, period: int):
        self.root, self.share, self.forest, self.top = root, share, forest, top
        self.chunk_len, self.period = chunk_len, period

    def free(self):
        for h in (self.forest, self.top):
            if h is not None and hasattr(h, "free"):
                h.free()
        self.forest = self.top = None


def commit_shares(engine, share: torch.Tensor, chunk_len: int, period: int, group=None) -> ShardedCommitment:
    roots, forest = engine.subtree_roots(share, chunk_len, keep=True)
    g = _all_gather(roots, group)                                     # [rank][q]: chunk q * world + rank of the vector
    digests = g.permute(1, 0, 2).reshape(-1, 32).contiguous()
    root, top = engine.root_of_digests(digests, keep=True)
    return ShardedCommitment(root, share, forest, top, chunk_len, period)


def open_shares(engine, cm: ShardedCommitment, positions: Sequence[int], group=None):
    world, rank = _world_rank(group)
    return open_in_shares(engine, cm.share, cm.forest, cm.top, cm.chunk_len, cm.period, positions, rank, world, group)


def stark_prove_sharded(engine, steps, stream, period: Optional[int] = None, group=None, timings: Optional[dict] = None
                        ) -> bytes:
    """`steps` runs the prover's steps off the FRI domain on this rank (the same on every rank):
      .fri_domain_length, .expansion_factor, .num_colinearity_tests, .offset, .omega
      .boundary_quotients() -> (registers, n, 2) coefficients, lowest first          stark.rs:237-288
      .randomizer()         -> (1, n, 2)   (after the transition quotients)          stark.rs:309-353
      .combination(seed)    -> (1, n, 2)   weights from seed = 32 transcript bytes   stark.rs:372-404
      .finish()             the prover's degree assertions                           stark.rs:379-380
    `stream` is this rank's copy of the proof stream (push(str), fiat_shamir(n) -> bytes, serialize() -> bytes,
    proof_stream.rs:23-58).  Returns the proof (stark.rs:470)."""
    import time
    world, rank = _world_rank(group)
    n, e, t = steps.fri_domain_length, steps.expansion_factor, steps.num_colinearity_tests
    offset, omega = steps.offset, steps.omega
    period = fri_shard_period(n, world, e, t, period)
    if period % world:
        raise ValueError("period must be a multiple of the number of ranks")
    c = period // world
    tm = {"steps": 0.0, "lde": 0.0, "merkle": 0.0, "fri": 0.0, "openings": 0.0}
    mark = [time.perf_counter()]

    def lap(slot):
        if timings is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            now = time.perf_counter()
            tm[slot] += (now - mark[0]) * 1e3
            mark[0] = now

    commits: List[ShardedCommitment] = []
    try:
        bq = steps.boundary_quotients()
        lap("steps")
        bq_shares = lde_to_shares(engine, bq, n, offset, period, group)                              # stark.rs:291-297
        lap("lde")
        for s in range(bq.shape[0]):
            commits.append(commit_shares(engine, bq_shares[s], c, period, group))                    # stark.rs:302-305
            stream.push(commits[-1].root.hex())
        lap("merkle")
        rz = steps.randomizer()
        lap("steps")
        rz_share = lde_to_shares(engine, rz, n, offset, period, group)[0]                            # stark.rs:356
        lap("lde")
        rz_commit = commit_shares(engine, rz_share, c, period, group)                                # stark.rs:358-360
        commits.append(rz_commit)
        stream.push(rz_commit.root.hex())
        lap("merkle")
        comb = steps.combination(stream.fiat_shamir(32))                                             # stark.rs:372-404
        lap("steps")
        comb_share = lde_to_shares(engine, comb, n, offset, period, group)[0]                        # stark.rs:406
        lap("lde")
        indices = sorted(fri_prove_sharded(engine, comb_share, n, offset, omega, e, t, stream, period, group))  # :409-410
        lap("fri")
        duplicated = sorted(indices + [(i + e) % n for i in indices])                                # stark.rs:415-419
        for cm in commits[:-1]:                                                                      # stark.rs:429-443
            for value, path in open_shares(engine, cm, duplicated, group):
                stream.push(json_field_element(value))
                stream.push(json_hash_path(path))
        for value, path in open_shares(engine, rz_commit, indices, group):                           # stark.rs:449-464
            stream.push(json_field_element(value))
            stream.push(json_hash_path(path))
        steps.finish()
        proof = stream.serialize()                                                                   # stark.rs:470
        lap("openings")
    finally:
        for cm in commits:
            cm.free()
    if timings is not None:
        tm["total"] = sum(tm.values())
        timings.update(tm)
    return proof

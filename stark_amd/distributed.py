"""Multi-GPU forms of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm) for the exchanges.

* Independent units (the columns of a batched coset LDE and their Merkle
  commitments, stark.rs:293-306 generalised) shard with NO data-path
  collective: column c goes to rank c mod G; only the 32-byte roots are
  gathered.
* One transform too large or too slow for one GPU uses a four-step
  decomposition n = n1 * n2 whose transposes are all-to-all exchanges.  The
  reference has no distributed code; the composition below equals
  NTT::forward / NTT::inverse (ntt.rs:25-40) on the concatenated vector.

The local arithmetic is delegated to an engine object so that the exchange
logic can be exercised on CPU tensors with the gloo backend."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------
# engines
# ---------------------------------------------------------------------------
class HipEngine:
    """Local arithmetic on the GPU through the C ABI (device pointers)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def ntt_batch(self, t: torch.Tensor, inverse: bool) -> None:
        cols, n, _ = t.shape
        self.ctx.ntt_dev(t, n, inverse=inverse, n_cols=cols, col_stride=n)

    def twiddle(self, t: torch.Tensor, n_total: int, row_start: int, inverse: bool) -> None:
        rows, cols, _ = t.shape
        self.ctx.fourstep_twiddle_dev(t, n_total, row_start, rows, cols, inverse=inverse)

    def coset_lde(self, coeffs: torch.Tensor, n_out: int, offset: int) -> torch.Tensor:
        cols, n_coeffs, _ = coeffs.shape
        out = torch.empty((cols, n_out, 2), dtype=coeffs.dtype, device=coeffs.device)
        self.ctx.coset_lde_batch_dev(coeffs, n_coeffs, n_coeffs, cols, n_out, offset, out, n_out)
        return out

    def merkle_root(self, column: torch.Tensor) -> bytes:
        return self.ctx.merkle_commit_fe_dev(column, column.shape[0])

    def merkle_roots(self, columns: torch.Tensor) -> List[bytes]:
        """Roots of all columns of a contiguous (cols, n, 2) batch in one call."""
        cols, n, _ = columns.shape
        return self.ctx.merkle_roots_fe_batch_dev(columns, n, cols, n)


# ---------------------------------------------------------------------------
# column sharding (no data-path collective)
# ---------------------------------------------------------------------------
def shard_columns(n_cols: int, rank: int, world: int) -> List[int]:
    """Columns owned by `rank`: c with c mod world == rank."""
    return list(range(rank, n_cols, world))


def gather_roots(local_roots: Sequence[bytes], n_cols: int, group=None) -> List[bytes]:
    """All ranks learn all column roots (32 bytes each), in column order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return list(local_roots)
    per_rank = (n_cols + world - 1) // world
    buf = torch.zeros((per_rank, 32), dtype=torch.uint8)
    for i, r in enumerate(local_roots):
        buf[i] = torch.frombuffer(bytearray(r), dtype=torch.uint8)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = buf.to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    roots: List[Optional[bytes]] = [None] * n_cols
    for r in range(world):
        host = out[r].cpu()
        for i, c in enumerate(shard_columns(n_cols, r, world)):
            roots[c] = bytes(host[i].tolist())
    assert rank >= 0 and all(x is not None for x in roots)
    return roots  # type: ignore[return-value]


def lde_commit_columns(engine, coeffs_local: torch.Tensor, n_cols: int, n_out: int, offset: int, group=None):
    """Batched coset LDE + per-column Merkle root, column-sharded.

    coeffs_local: this rank's columns (shard_columns order), shape
    (local_cols, n_coeffs, 2) int64.  Returns (codewords_local, all_roots)."""
    cw = engine.coset_lde(coeffs_local, n_out, offset)
    if hasattr(engine, "merkle_roots") and cw.shape[0] > 0:
        local_roots = engine.merkle_roots(cw)
    else:
        local_roots = [engine.merkle_root(cw[i]) for i in range(cw.shape[0])]
    return cw, gather_roots(local_roots, n_cols, group)


# ---------------------------------------------------------------------------
# four-step NTT across ranks
# ---------------------------------------------------------------------------
def _a2a_transpose(t: torch.Tensor, group) -> torch.Tensor:
    """t is this rank's rows of a (A x B) matrix, shape (A/G, B, 2).  Returns
    this rank's rows of the transposed (B x A) matrix, shape (B/G, A, 2).
    One all-to-all; each rank sends (A/G)*(B/G) elements to every peer."""
    world = dist.get_world_size(group)
    a_loc, b, _ = t.shape
    assert b % world == 0
    send = t.view(a_loc, world, b // world, 2).permute(1, 2, 0, 3).contiguous()      # [dest][b_dest][a_loc]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)                # [src][b_loc][a_src]
    return recv.view(world, b // world, a_loc, 2).permute(1, 0, 2, 3).reshape(b // world, world * a_loc, 2)


def fourstep_split(n: int, world: int):
    log_n = n.bit_length() - 1
    assert n == 1 << log_n, "NTT size must be power of 2"
    n1 = 1 << ((log_n + 1) // 2)
    n2 = n // n1
    assert n1 % world == 0 and n2 % world == 0, "transform too small for this many ranks"
    return n1, n2


def fourstep_ntt(x_local: torch.Tensor, n: int, engine, inverse: bool = False, natural_output: bool = True,
                 group=None) -> torch.Tensor:
    """Size-n transform of the vector whose contiguous slice
    [rank*n/G, (rank+1)*n/G) is x_local (shape (n/G, 2), int64 {lo,hi}).

    View the input as an n1 x n2 row-major matrix x[j1][j2].  Then
      X[k1 + n1*k2] = sum_j2 w_n^(j2*k1) w_n2^(j2*k2) sum_j1 x[j1][j2] w_n1^(j1*k1):
    transpose (all-to-all), n2/G column transforms of size n1, twiddle,
    transpose, n1/G transforms of size n2, and a last transpose that puts the
    result back in natural order (rank r holds X[r*n/G : (r+1)*n/G]).  With
    natural_output=False the last exchange is skipped and the result is left
    as rows k1 (this rank's block) x columns k2."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n1, n2 = fourstep_split(n, world)
    assert x_local.shape == (n // world, 2)
    t = _a2a_transpose(x_local.view(n1 // world, n2, 2), group)          # [j2_local][j1]
    engine.ntt_batch(t, inverse)                                          # [j2_local][k1]
    engine.twiddle(t, n, rank * (n2 // world), inverse)                   # * w_n^(j2*k1)
    t = _a2a_transpose(t, group)                                          # [k1_local][j2]
    engine.ntt_batch(t, inverse)                                          # [k1_local][k2]
    if not natural_output:
        return t
    t = _a2a_transpose(t, group)                                          # [k2_local][k1]
    return t.reshape(n // world, 2)

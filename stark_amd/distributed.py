"""Multi-GPU forms of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm) for the exchanges.

* Independent units (the columns of a batched coset LDE and their Merkle
  commitments, stark.rs:293-306 generalised) shard with NO data-path
  collective: column c goes to rank c mod G; only the 32-byte roots are
  gathered.
* The commit phase of FRI (fri.rs:100-152) on one large codeword deals every
  layer out in chunks so that position i and its fold partner i + N/2 live on
  the same rank: folds and leaf hashing are local, each rank reduces its chunks
  to subtree roots, and only those 32-byte roots are all-gathered per round.
  Once a layer is down to one chunk per rank it is all-gathered whole and the
  remaining (latency-bound) rounds run on every rank alike.
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

    def _bind(self, t: torch.Tensor) -> None:
        """The library launches on the context's stream; the tensors it is handed were produced by
        torch ops and collectives on torch's current stream, and its results are consumed there.
        Every engine call therefore first points the context at torch's current stream for the
        tensor's device (it may have changed under `with torch.cuda.stream(s)`), which orders the
        two by plain stream order.  Switching synchronises the stream the context leaves.  The
        context itself remembers which stream it is on (Context.follow_stream), so a set_stream made
        elsewhere on the same Context is seen here."""
        if not t.is_cuda:
            return
        self.ctx.follow_stream(torch.cuda.current_stream(t.device).cuda_stream)

    def ntt_batch(self, t: torch.Tensor, inverse: bool) -> None:
        self._bind(t)
        cols, n, _ = t.shape
        self.ctx.ntt_dev(t, n, inverse=inverse, n_cols=cols, col_stride=n)

    def twiddle(self, t: torch.Tensor, n_total: int, row_start: int, inverse: bool) -> None:
        self._bind(t)
        rows, cols, _ = t.shape
        self.ctx.fourstep_twiddle_dev(t, n_total, row_start, rows, cols, inverse=inverse)

    def pack(self, t: torch.Tensor, world: int, twiddle=None) -> torch.Tensor:
        """Send buffer [dest][column within dest's share][row] of this rank's rows t (rows, cols, 2), with the
        four-step twiddle (n_total, row_start, inverse) applied on the way if given: one HIP kernel."""
        self._bind(t)
        rows, cols, _ = t.shape
        send = torch.empty((world, cols // world, rows, 2), dtype=t.dtype, device=t.device)
        if twiddle is None:
            self.ctx.fourstep_pack_dev(t, rows, cols, world, send)
        else:
            n_total, row_start, inverse = twiddle
            self.ctx.fourstep_pack_dev(t, rows, cols, world, send, n_total, row_start, -1 if inverse else 1)
        return send

    def unpack(self, recv: torch.Tensor, world: int, rows: int, run: int) -> torch.Tensor:
        """Received blocks [source][row][run] -> rows (rows, world * run, 2): one HIP kernel."""
        self._bind(recv)
        out = torch.empty((rows, world * run, 2), dtype=recv.dtype, device=recv.device)
        self.ctx.fourstep_unpack_dev(recv, world, rows, run, out)
        return out

    def compose_linear(self, t: torch.Tensor, s: int) -> None:
        """Coefficient j (lowest first) of every column of t (cols, n, 2) times s^j, in place."""
        self._bind(t)
        cols, n, _ = t.shape
        self.ctx.poly_compose_linear_dev(t, n, s, n_cols=cols, col_stride=n)

    def transpose(self, t: torch.Tensor) -> torch.Tensor:
        """(rows, cols, 2) -> (cols, rows, 2): the pack kernel with one destination."""
        self._bind(t)
        rows, cols, _ = t.shape
        out = torch.empty((cols, rows, 2), dtype=t.dtype, device=t.device)
        self.ctx.fourstep_pack_dev(t, rows, cols, 1, out)
        return out

    def coset_lde(self, coeffs: torch.Tensor, n_out: int, offset: int) -> torch.Tensor:
        self._bind(coeffs)
        cols, n_coeffs, _ = coeffs.shape
        out = torch.empty((cols, n_out, 2), dtype=coeffs.dtype, device=coeffs.device)
        self.ctx.coset_lde_batch_dev(coeffs, n_coeffs, n_coeffs, cols, n_out, offset, out, n_out)
        return out

    def merkle_root(self, column: torch.Tensor) -> bytes:
        self._bind(column)
        return self.ctx.merkle_commit_fe_dev(column, column.shape[0])

    def merkle_roots(self, columns: torch.Tensor) -> List[bytes]:
        """Roots of all columns of a contiguous (cols, n, 2) batch in one call."""
        self._bind(columns)
        cols, n, _ = columns.shape
        return self.ctx.merkle_roots_fe_batch_dev(columns, n, cols, n)

    # ---- sharded FRI commit ------------------------------------------------
    def subtree_roots(self, local: torch.Tensor, chunk_len: int, keep: bool = False):
        """(roots (n_chunks, 32) on the device, forest handle or None)."""
        self._bind(local)
        n_chunks = local.shape[0] // chunk_len
        roots = torch.empty((n_chunks, 32), dtype=torch.uint8, device=local.device)
        forest = self.ctx.merkle_subtree_roots_dev(local, chunk_len, n_chunks, roots, keep_forest=keep)
        return roots, forest

    def root_of_digests(self, digests: torch.Tensor, keep: bool = False):
        """root, or (root, tree handle) with keep."""
        self._bind(digests)
        return self.ctx.merkle_root_of_digests_dev(digests, digests.shape[0], keep_tree=keep)

    def fold_shard(self, local: torch.Tensor, alpha: int, offset: int, omega: int, chunk_len: int, period: int,
                   base: int, want_roots: bool, keep: bool = False):
        """(folded share, its chunk roots or None, forest handle or None)."""
        self._bind(local)
        half = local.shape[0] // 2
        out = torch.empty((half, 2), dtype=local.dtype, device=local.device)
        roots = torch.empty((half // chunk_len, 32), dtype=torch.uint8, device=local.device) if want_roots else None
        forest = self.ctx.fri_fold_shard_dev(local, local.shape[0], alpha, offset, omega, chunk_len, period, base,
                                             out, roots, keep_forest=keep and want_roots)
        return out, roots, forest

    def read_elements(self, share: torch.Tensor, local_indices: Sequence[int]) -> List[int]:
        """Values (as integers) of a few elements of a resident share."""
        if not local_indices:
            return []
        idx = torch.tensor(list(local_indices), dtype=torch.int64, device=share.device)
        rows = share[idx].cpu().tolist()
        return [(lo & 0xFFFFFFFFFFFFFFFF) | ((hi & 0xFFFFFFFFFFFFFFFF) << 64) for lo, hi in rows]

    def fri_commit(self, codeword: torch.Tensor, offset: int, omega: int, expansion_factor: int,
                   num_colinearity_tests: int, transcript):
        self._bind(codeword)
        return self.ctx.fri_commit(codeword, offset, omega, expansion_factor, num_colinearity_tests, transcript,
                                   n=codeword.shape[0])


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
def _all_to_all(send: torch.Tensor, group) -> torch.Tensor:
    """Equal-split all-to-all of a flat send buffer [dest][...]; through the host when the group's backend
    cannot take device tensors (gloo in tests)."""
    if dist.get_backend(group) == "nccl" or not send.is_cuda:
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
        return recv
    host = send.cpu()
    out = torch.empty_like(host)
    dist.all_to_all_single(out.view(-1), host.view(-1), group=group)
    return out.to(send.device)


def _a2a_transpose(t: torch.Tensor, group, engine=None, twiddle=None) -> torch.Tensor:
    """t is this rank's rows of a (A x B) matrix, shape (A/G, B, 2).  Returns
    this rank's rows of the transposed (B x A) matrix, shape (B/G, A, 2).
    One all-to-all; each rank sends (A/G)*(B/G) elements to every peer.
    twiddle = (n_total, row_start, inverse): element (row, column) is multiplied by
    w_n^((row_start + row) * column) before it travels.  An engine with pack/unpack
    (the HIP one) does the local rearrangement and the twiddle in its own kernels;
    without them (CPU tensors in tests) torch moves the data and engine.twiddle
    multiplies."""
    world = dist.get_world_size(group)
    a_loc, b, _ = t.shape
    assert b % world == 0
    if engine is not None and hasattr(engine, "pack") and t.is_cuda:
        send = engine.pack(t, world, twiddle)                                        # [dest][b_dest][a_loc]
        recv = _all_to_all(send, group)                                              # [src][b_loc][a_src]
        return engine.unpack(recv, world, b // world, a_loc)                         # [b_loc][src][a_src]
    if twiddle is not None:
        engine.twiddle(t, *twiddle)
    send = t.view(a_loc, world, b // world, 2).permute(1, 2, 0, 3).contiguous()
    recv = _all_to_all(send, group)
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
    t = _a2a_transpose(x_local.view(n1 // world, n2, 2), group, engine)  # [j2_local][j1]
    engine.ntt_batch(t, inverse)                                          # [j2_local][k1]
    t = _a2a_transpose(t, group, engine,                                  # * w_n^(j2*k1), then [k1_local][j2]
                       twiddle=(n, rank * (n2 // world), inverse))
    engine.ntt_batch(t, inverse)                                          # [k1_local][k2]
    if not natural_output:
        return t
    t = _a2a_transpose(t, group, engine)                                  # [k2_local][k1]
    return t.reshape(n // world, 2)


# ---------------------------------------------------------------------------
# FRI commit phase across ranks
# ---------------------------------------------------------------------------
P = 407 * (1 << 119) + 1          # field.rs:32


def fri_num_rounds(domain_length: int, expansion_factor: int, num_colinearity_tests: int) -> int:
    """Fri::num_rounds (fri.rs:39-51)."""
    length, rounds = domain_length, 0
    while length > expansion_factor and 4 * num_colinearity_tests < length:
        length //= 2
        rounds += 1
    return rounds


def fri_shard_period(n: int, world: int, expansion_factor: int, num_colinearity_tests: int,
                     period: Optional[int] = None) -> int:
    """The period of the chunked layout: layers longer than it are folded
    shard-wise.  At least one chunk element per rank, at most n, and never so
    small that a sharded fold would be the last round's."""
    rounds = fri_num_rounds(n, expansion_factor, num_colinearity_tests)
    if rounds < 1:
        raise ValueError("Cannot do FRI with less than 1 round")
    if period is None:
        period = 1 << 17
    period = max(period, world, n >> (rounds - 1))
    return min(period, n)


def fri_shard_of(codeword: torch.Tensor, rank: int, world: int, period: int) -> torch.Tensor:
    """This rank's share of a whole layer (n, 2): chunk `rank` of every period,
    back to back; local element q*C + j is position q*period + rank*C + j,
    C = period / world."""
    n = codeword.shape[0]
    c = period // world
    return codeword.view(n // period, world, c, 2)[:, rank].reshape(n // period * c, 2).contiguous()


def _all_gather(t: torch.Tensor, group) -> torch.Tensor:
    """(world, *t.shape), on t's device; through the host when the group's backend cannot take device tensors."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return t.unsqueeze(0)
    src = t.contiguous()
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=group)
        return out
    if src.is_cuda:
        src = src.cpu()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return torch.stack(parts).to(t.device)


class ShardedFriCommit:
    """Result of fri_commit_sharded: `roots` of all rounds; `shares[k]`, this
    rank's share of layer k for the layers folded shard-wise; `tail`, the
    engine's own commit object for the layers from `len(shares)` on, whole on
    every rank.  With keep_trees also `forests[k]` (this rank's subtrees of
    layer k) and `tops[k]` (the levels above the subtree roots, on every rank)."""

    def __init__(self, roots, shares, tail, period, chunk_len, n, forests=None, tops=None):
        self.roots: List[bytes] = roots
        self.shares: List[torch.Tensor] = shares
        self.tail = tail
        self.period = period
        self.chunk_len = chunk_len
        self.n = n
        self.forests = forests or []
        self.tops = tops or []

    def free(self):
        for h in self.forests + self.tops + [self.tail]:
            if h is not None and hasattr(h, "free"):
                h.free()
        self.forests, self.tops = [], []

    def gather_layer(self, k: int, group=None) -> torch.Tensor:
        """Layer k < len(shares) in natural order on every rank."""
        g = _all_gather(self.shares[k], group)                          # [rank][q][j]
        world = g.shape[0]
        return g.view(world, -1, self.chunk_len, 2).permute(1, 0, 2, 3).reshape(-1, 2)


def fri_commit_sharded(engine, local: torch.Tensor, n: int, offset: int, omega: int, expansion_factor: int,
                       num_colinearity_tests: int, transcript, period: Optional[int] = None,
                       group=None, keep_trees: bool = False) -> ShardedFriCommit:
    """Fri::commit (fri.rs:100-152) of a length-n codeword of which this rank
    holds `local` = fri_shard_of(codeword, rank, world, period).

    transcript(round, last, root) -> alpha is called on EVERY rank with the same
    arguments (each rank keeps its own copy of the proof stream; the roots are
    identical, so the challenges agree without a broadcast).  Per sharded round
    the only exchange is an all-gather of 32 bytes per local chunk."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if n < 1 or n & (n - 1):
        raise ValueError("length must be power of two")
    if world & (world - 1):
        raise ValueError("the chunked layout needs a power-of-two number of ranks")
    period = fri_shard_period(n, world, expansion_factor, num_colinearity_tests, period)
    if period & (period - 1) or period % world:
        raise ValueError("period must be a power of two and a multiple of the number of ranks")
    c = period // world
    if local.shape != (n // period * c, 2):
        raise ValueError("local share does not match the chunked layout")
    sharded_rounds = (n // period).bit_length() - 1

    roots: List[bytes] = []
    shares: List[torch.Tensor] = []
    forests, tops = [], []
    om, of = omega % P, offset % P
    chunk_roots, forest = engine.subtree_roots(local, c, keep=keep_trees) if sharded_rounds else (None, None)
    for k in range(sharded_rounds):
        g = _all_gather(chunk_roots, group)                             # [rank][q]; chunk q*world + rank of the layer
        digests = g.permute(1, 0, 2).reshape(-1, 32).contiguous()
        if keep_trees:
            root, top = engine.root_of_digests(digests, keep=True)      # fri.rs:119
            forests.append(forest)
            tops.append(top)
        else:
            root = engine.root_of_digests(digests)
        roots.append(root)
        alpha = transcript(k, False, root)                              # fri.rs:120,126-127
        shares.append(local)
        local, chunk_roots, forest = engine.fold_shard(local, alpha, of, om, c, period, rank * c,
                                                       want_roots=k + 1 < sharded_rounds, keep=keep_trees)  # fri.rs:133-139
        om, of = om * om % P, of * of % P                               # fri.rs:141-142
    whole = _all_gather(local, group).reshape(period, 2)                # one chunk per rank: the layer in order
    tail = engine.fri_commit(whole, of, om, expansion_factor, num_colinearity_tests,
                             lambda r, last, root: transcript(sharded_rounds + r, last, root))
    roots.extend(tail.root(i) for i in range(len(tail)))
    return ShardedFriCommit(roots, shares, tail, period, c, n, forests, tops)


# ---- the query phase on the sharded layers ----------------------------------
def _u32_digits(v: int) -> List[int]:
    d = []
    while v:
        d.append(v & 0xFFFFFFFF)
        v >>= 32
    return d


def json_field_vec(values: Sequence[int]) -> str:
    """serde_json::to_string(&Vec<FieldElement>) (fri.rs:146): BigInt as [sign, u32 digits low first]."""
    return "[" + ",".join('{"value":[%d,[%s]]}' % (0 if v == 0 else 1, ",".join(map(str, _u32_digits(v))))
                          for v in values) + "]"


def json_hash_path(path: Sequence[bytes]) -> str:
    """serde_json::to_string(&Vec<HashOutput>) (fri.rs:188)."""
    return "[" + ",".join("[" + ",".join(str(b) for b in h) + "]" for h in path) + "]"


def fri_sample_indices(seed: bytes, size: int, reduced_size: int, number: int) -> List[int]:
    """Fri::sample_indices / sample_index (fri.rs:54-87): BLAKE2b-512(seed || counter u64 LE), folded
    big-endian into a usize (the last eight bytes survive), mod size; distinct modulo reduced_size."""
    import hashlib
    indices: List[int] = []
    reduced: List[int] = []
    counter = 0
    while len(indices) < number:
        h = hashlib.blake2b(seed + counter.to_bytes(8, "little"), digest_size=64).digest()
        index = int.from_bytes(h[-8:], "big") % size
        counter += 1
        if index % reduced_size not in reduced:
            indices.append(index)
            reduced.append(index % reduced_size)
    return indices


def _open_many(handle, indices: Sequence[int]) -> List[List[bytes]]:
    """Paths of several indices from a retained tree or forest: one launch when the handle offers it."""
    if not indices:
        return []
    if hasattr(handle, "open_many"):
        return handle.open_many(list(indices))
    return [handle.open(i) for i in indices]


def open_in_shares(engine, share: torch.Tensor, forest, top, chunk_len: int, period: int, positions: Sequence[int],
                   rank: int, world: int, group):
    """Values and authentication paths (Merkle::open, merkle.rs:54-93) of `positions` of a vector dealt out in
    chunks (fri_shard_of), on every rank: the owner of a position reads the element and the siblings inside
    its subtree (`forest`); those travel in one zero-padded all-reduce (non-owners contribute zeros); the
    upper siblings come from the replicated top levels (`top`)."""
    c = chunk_len
    sub_levels = c.bit_length() - 1
    slot = 16 + 32 * sub_levels
    buf = torch.zeros((len(positions), slot), dtype=torch.uint8)
    mine = [(s, (i // period) * c + i % c) for s, i in enumerate(positions) if (i % period) // c == rank]
    vals = engine.read_elements(share, [l for _, l in mine])
    subs = _open_many(forest, [l for _, l in mine])
    for (s, _), v, sub in zip(mine, vals, subs):
        raw = v.to_bytes(16, "little") + b"".join(sub)
        buf[s] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else buf.device
        t = buf.to(torch.int32).to(dev)                  # byte sums of one owner and zeros: exact in any integer type
        dist.all_reduce(t, group=group)
        buf = t.cpu().to(torch.uint8)
    uppers = _open_many(top, [i // c for i in positions])
    out = []
    for s in range(len(positions)):
        raw = bytes(buf[s].tolist())
        value = int.from_bytes(raw[:16], "little")
        out.append((value, [raw[16 + 32 * j:48 + 32 * j] for j in range(sub_levels)] + uppers[s]))
    return out


def _sharded_openings(engine, commit: ShardedFriCommit, layer: int, positions: Sequence[int], rank: int, world: int,
                      group):
    return open_in_shares(engine, commit.shares[layer], commit.forests[layer], commit.tops[layer], commit.chunk_len,
                          commit.period, positions, rank, world, group)


def fri_prove_sharded(engine, local: torch.Tensor, n: int, offset: int, omega: int, expansion_factor: int,
                      num_colinearity_tests: int, stream, period: Optional[int] = None, group=None) -> List[int]:
    """Fri::prove (fri.rs:212-254) of a codeword dealt out over the ranks as in
    fri_commit_sharded.  `stream` is this rank's copy of the proof stream
    (push(str), fiat_shamir(nbytes) -> bytes, as proof_stream.rs:23-58); every
    rank ends with the same stream, byte for byte the single-GPU one.  Returns
    the first round's a and b indices."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    t = num_colinearity_tests

    def transcript(rnd, last, root):
        stream.push(root.hex())                                         # fri.rs:120
        return None if last else int.from_bytes(stream.fiat_shamir(32), "big") % P   # fri.rs:126-127, field.rs:110-116

    commit = fri_commit_sharded(engine, local, n, offset, omega, expansion_factor, t, transcript, period, group,
                                keep_trees=True)
    try:
        k_sh, tail = len(commit.shares), commit.tail
        rounds = k_sh + len(tail)
        last_len = n >> (rounds - 1)
        last_raw = tail.download(len(tail) - 1)
        stream.push(json_field_vec([int(lo) | (int(hi) << 64) for lo, hi in last_raw.tolist()]))    # fri.rs:146
        top = fri_sample_indices(stream.fiat_shamir(32), n // 2, last_len, t)                      # fri.rs:218-223

        def openings(layer, positions):
            if layer < k_sh:
                return _sharded_openings(engine, commit, layer, positions, rank, world, group)
            vals = tail.read(layer - k_sh, positions)
            paths = tail.open_many(layer - k_sh, positions) if hasattr(tail, "open_many") else \
                [tail.open(layer - k_sh, i) for i in positions]
            return [(int(lo) | (int(hi) << 64), p) for (lo, hi), p in zip(vals.tolist(), paths)]

        indices = list(top)
        for i in range(rounds - 1):                                     # fri.rs:228-242
            half = (n >> i) // 2
            cs = [idx % half for idx in indices][:t]
            a, b = cs, [x + half for x in cs]
            cur = openings(i, a + b)
            nxt = openings(i + 1, cs)
            for s in range(t):                                          # fri.rs:166-179
                stream.push('["%d","%d","%d"]' % (cur[s][0], cur[t + s][0], nxt[s][0]))
            for s in range(t):                                          # fri.rs:182-204
                stream.push(json_hash_path(cur[s][1]))
                stream.push(json_hash_path(cur[t + s][1]))
                stream.push(json_hash_path(nxt[s][1]))
            indices = a + b
        return list(top) + [x + n // 2 for x in top]                    # fri.rs:246-253
    finally:
        commit.free()

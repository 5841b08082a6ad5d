"""world_size-2 (and 4) gloo runs of the multi-GPU exchange logic on CPU: the
local arithmetic is the C restatement, the all-to-all / gather code is the
product's (stark_amd/distributed.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in for HipEngine built on the C restatement."""

    def __init__(self):
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import oracle_c
        import stark_oracle
        self.oc, self.so = oracle_c, stark_oracle

    def ntt_batch(self, t, inverse):
        a = t.numpy().view(np.uint64)
        for c in range(a.shape[0]):
            a[c] = self.oc.ntt(np.ascontiguousarray(a[c]), inverse=inverse)

    def twiddle(self, t, n_total, row_start, inverse):
        so, oc = self.so, self.oc
        w = so.primitive_nth_root(n_total)
        if inverse:
            w = so.fe_inverse(w)
        a = t.numpy().view(np.uint64)
        rows, cols, _ = a.shape
        for i in range(rows):
            base = so.fe_pow(w, row_start + i)
            tw = [1] * cols
            for k in range(1, cols):
                tw[k] = so.fe_mul(tw[k - 1], base)
            out = np.empty((cols, 2), dtype=np.uint64)
            oc.lib.so_fe_mul_vec(oc.p64(np.ascontiguousarray(a[i])), oc.p64(oc.from_ints(tw)), oc.p64(out), cols)
            a[i] = out

    def coset_lde(self, coeffs, n_out, offset):
        a = coeffs.numpy().view(np.uint64)
        out = np.stack([self.oc.coset_lde(np.ascontiguousarray(a[c]), n_out, offset) for c in range(a.shape[0])]) \
            if a.shape[0] else np.empty((0, n_out, 2), dtype=np.uint64)
        return torch.from_numpy(out.view(np.int64))

    def merkle_root(self, column):
        return self.oc.merkle_commit_fe(np.ascontiguousarray(column.numpy().view(np.uint64)))

    # ---- sharded FRI commit / prove ----------------------------------------
    class _Levels:
        """Levels of digests (leaf level first) with Merkle::open's sibling walk (merkle.rs:54-93)."""

        def __init__(self, levels, n_levels):
            self.levels, self.n_levels = levels, n_levels

        def open(self, index):
            return [self.levels[k][(index >> k) ^ 1] for k in range(self.n_levels)]

        def free(self):
            pass

    def _reduce(self, level, n_levels):
        levels = [level]
        for _ in range(n_levels):                              # merkle.rs:38-51 pairs (2i, 2i+1)
            level = [self.oc.blake2b(level[i] + level[i + 1]) for i in range(0, len(level), 2)]
            levels.append(level)
        return levels

    def subtree_roots(self, local, chunk_len, keep=False):
        a = self.oc.to_ints(local.numpy().view(np.uint64))
        leaves = [self.oc.blake2b(self.oc.bincode_fe(v)) for v in a]
        n_levels = chunk_len.bit_length() - 1
        levels = self._reduce(leaves, n_levels)
        roots = torch.from_numpy(np.frombuffer(b"".join(levels[-1]), dtype=np.uint8).reshape(-1, 32).copy())
        return roots, (self._Levels(levels, n_levels) if keep else None)

    def root_of_digests(self, digests, keep=False):
        level = [bytes(d.tolist()) for d in digests]
        n_levels = len(level).bit_length() - 1
        levels = self._reduce(level, n_levels)
        return (levels[-1][0], self._Levels(levels, n_levels)) if keep else levels[-1][0]

    def fold_shard(self, local, alpha, offset, omega, chunk_len, period, base, want_roots, keep=False):
        so, oc = self.so, self.oc
        a = oc.to_ints(local.numpy().view(np.uint64))
        half = len(a) // 2
        two_inv = so.fe_inverse(2)
        out = []
        for l in range(half):                                  # fri.rs:133-138 at the element's position in the layer
            pos = (l // chunk_len) * period + base + l % chunk_len
            r = so.fe_div(alpha, so.fe_mul(offset, so.fe_pow(omega, pos)))
            out.append(so.fe_mul(two_inv, so.fe_add(so.fe_mul(so.fe_add(1, r), a[l]),
                                                    so.fe_mul(so.fe_sub(1, r), a[half + l]))))
        t = torch.from_numpy(oc.from_ints(out).view(np.int64))
        roots, forest = self.subtree_roots(t, chunk_len, keep) if want_roots else (None, None)
        return t, roots, forest

    def read_elements(self, share, local_indices):
        a = share.numpy().view(np.uint64)
        return [self.oc.to_int(a[l]) for l in local_indices]

    def fri_commit(self, codeword, offset, omega, expansion, ncol, transcript):
        so, oc = self.so, self.oc
        cw = np.ascontiguousarray(codeword.numpy().view(np.uint64))
        rounds = oc.lib.so_fri_num_rounds(cw.shape[0], expansion, ncol)
        layers, roots, trees = [], [], []
        for r in range(rounds):
            trees.append(oc.merkle_tree_fe(cw))
            root = oc.merkle_commit_fe(cw)
            roots.append(root)
            layers.append(cw)
            alpha = transcript(r, r == rounds - 1, root)
            if r == rounds - 1:
                break
            cw = oc.fri_fold(cw, alpha, offset, omega)
            offset, omega = so.fe_mul(offset, offset), so.fe_mul(omega, omega)

        class Tail:
            def __len__(self):
                return len(roots)

            def root(self, i):
                return roots[i]

            def download(self, i):
                return layers[i]

            def read(self, i, idx):
                return layers[i][list(idx)]

            def open(self, i, index):
                return oc.merkle_open(trees[i], index)

            def free(self):
                pass
        tail = Tail()
        tail.layers = layers
        return tail


def _fri_worker(rank, world, port, log_n, period, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from stark_amd import distributed as D
        eng = OracleEngine()
        oc, so = eng.oc, eng.so
        n, expansion, ncol = 1 << log_n, 4, 2
        x = oc.splitmix(n, 123)
        omega, offset = so.primitive_nth_root(n), so.GENERATOR
        ref_layers, ref_roots, ref_alphas, _ = oc.fri_commit(x, offset, omega, expansion, ncol)
        used = D.fri_shard_period(n, world, expansion, ncol, period)
        local = D.fri_shard_of(torch.from_numpy(x.view(np.int64)), rank, world, used)
        stream, alphas = oc.Stream(), []

        def transcript(rnd, last, root):
            stream.push(root.hex())
            if last:
                return None
            alphas.append(so.fe_sample(stream.fiat_shamir(32)))
            return alphas[-1]

        res = D.fri_commit_sharded(eng, local, n, offset, omega, expansion, ncol, transcript, period=period)
        ok = res.roots == ref_roots and alphas == ref_alphas
        k_sh = len(res.shares)
        for k in range(k_sh):
            ok = ok and np.array_equal(res.gather_layer(k).numpy().view(np.uint64), ref_layers[k])
        for i, lay in enumerate(res.tail.layers):
            ok = ok and np.array_equal(lay, ref_layers[k_sh + i])
        ok = ok and k_sh + len(res.tail.layers) == len(ref_layers)
        q.put((rank, ok, k_sh))
    finally:
        dist.destroy_process_group()


def _fri_prove_worker(rank, world, port, log_n, period, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from stark_amd import distributed as D
        eng = OracleEngine()
        oc, so = eng.oc, eng.so
        n, expansion, ncol = 1 << log_n, 4, 3
        # a codeword FRI accepts: evaluations of a polynomial of degree < n/expansion on the coset
        coeffs = oc.splitmix(n // expansion, 321)
        omega, offset = so.primitive_nth_root(n), so.GENERATOR
        x = oc.coset_lde(coeffs, n, offset)
        ref_stream = so.ProofStream()
        fri = so.Fri(offset, omega, n, expansion, ncol)
        ref_indices = fri.prove(oc.to_ints(x), ref_stream)
        used = D.fri_shard_period(n, world, expansion, ncol, period)
        local = D.fri_shard_of(torch.from_numpy(x.view(np.int64)), rank, world, used)

        class Stream:                                          # proof_stream.rs:23-58
            def __init__(self):
                self.ps = so.ProofStream()

            def push(self, s):
                self.ps.push(s)

            def fiat_shamir(self, nbytes):
                return self.ps.prover_fiat_shamir(nbytes)
        stream = Stream()
        indices = D.fri_prove_sharded(eng, local, n, offset, omega, expansion, ncol, stream, period=period)
        same = stream.ps.objects == ref_stream.objects and indices == ref_indices
        # and the reference verifier's restatement accepts the stream
        ver = so.ProofStream.deserialize(stream.ps.serialize())
        accepted = fri.verify(ver, [])
        q.put((rank, bool(same), bool(accepted)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,period", [(2, 9, 32), (4, 10, 16), (2, 7, None)])
def test_fri_prove_sharded_gloo(world, log_n, period):
    """The proof stream of the sharded Fri::prove equals the single-vector one
    object for object (roots, last codeword, colinearity triples, paths put
    together from an owner's subtree and the shared top levels), on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world * 13 + log_n
    procs = [ctx.Process(target=_fri_prove_worker, args=(r, world, port, log_n, period, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, accepted in results:
        assert same, f"rank {rank}: proof stream differs from the single-vector prover's"
        assert accepted, f"rank {rank}: verifier rejected"


@pytest.mark.parametrize("world,log_n,period,sharded", [(2, 10, 64, 4), (2, 9, None, 0), (4, 10, 4, 6), (2, 8, 2, 4)])
def test_fri_commit_sharded_gloo(world, log_n, period, sharded):
    """Every round's root, every challenge and every layer equal the
    single-vector commit; `sharded` = rounds folded shard-wise (a period below
    n >> (rounds-1) is raised to it so the last round is never a sharded one)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 11 + log_n
    procs = [ctx.Process(target=_fri_worker, args=(r, world, port, log_n, period, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, k_sh in results:
        assert ok, f"rank {rank}: sharded commit differs from the single-vector commit"
        assert k_sh == sharded


def _worker(rank, world, port, log_n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from stark_amd import distributed as D
        eng = OracleEngine()
        oc, so = eng.oc, eng.so
        n = 1 << log_n
        x = oc.splitmix(n, 99)
        per = n // world
        x_local = torch.from_numpy(x[rank * per:(rank + 1) * per].copy().view(np.int64))
        # forward, natural order
        y_local = D.fourstep_ntt(x_local, n, eng, inverse=False)
        ref = oc.ntt(x)
        ok_fwd = np.array_equal(y_local.numpy().view(np.uint64), ref[rank * per:(rank + 1) * per])
        # inverse of the forward result gives the input back
        z_local = D.fourstep_ntt(y_local.contiguous(), n, eng, inverse=True)
        ok_inv = np.array_equal(z_local.numpy().view(np.uint64), x[rank * per:(rank + 1) * per])
        # transposed-sharded output: rows k1 (this rank's block), columns k2, X[k1 + n1*k2]
        n1, n2 = D.fourstep_split(n, world)
        t = D.fourstep_ntt(x_local, n, eng, inverse=False, natural_output=False).numpy().view(np.uint64)
        k1 = np.arange(rank * (n1 // world), (rank + 1) * (n1 // world))
        idx = k1[:, None] + n1 * np.arange(n2)[None, :]
        ok_t = np.array_equal(t, ref[idx])
        # column-sharded LDE + roots
        n_cols, n_coeffs, n_out = 5, 64, 256
        coeffs = oc.splitmix(n_cols * n_coeffs, 5).reshape(n_cols, n_coeffs, 2)
        mine = D.shard_columns(n_cols, rank, world)
        local = torch.from_numpy(np.ascontiguousarray(coeffs[mine]).view(np.int64)) if mine else \
            torch.empty((0, n_coeffs, 2), dtype=torch.int64)
        cw, roots = D.lde_commit_columns(eng, local, n_cols, n_out, so.GENERATOR)
        ref_roots = [oc.merkle_commit_fe(oc.coset_lde(coeffs[c], n_out, so.GENERATOR)) for c in range(n_cols)]
        ok_roots = roots == ref_roots
        q.put((rank, ok_fwd, ok_inv, ok_t, ok_roots))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 10), (2, 13), (4, 12)])
def test_fourstep_and_column_sharding_gloo(world, log_n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 7 + log_n
    procs = [ctx.Process(target=_worker, args=(r, world, port, log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_fwd, ok_inv, ok_t, ok_roots in results:
        assert ok_fwd, f"rank {rank}: forward four-step != single-vector NTT"
        assert ok_inv, f"rank {rank}: inverse four-step did not round-trip"
        assert ok_t, f"rank {rank}: transposed-sharded layout mismatch"
        assert ok_roots, f"rank {rank}: column roots mismatch"


def test_shard_columns():
    from stark_amd.distributed import shard_columns
    assert shard_columns(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((shard_columns(5, r, 4) for r in range(4)), [])) == list(range(5))


def test_prove_helpers_match_restatement():
    """The wire strings and the index sampling the sharded prover uses are the
    restatement's (fri.rs:54-87,146,188), including the short encodings."""
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import stark_oracle as so
    from stark_amd import distributed as D
    vals = [0, 1, (1 << 32) - 1, 1 << 32, (1 << 64) + 5, 1 << 96, so.P - 1]
    assert D.json_field_vec(vals) == so.json_field_vec(vals)
    path = [bytes(range(32)), bytes([255] * 32), bytes(32)]
    assert D.json_hash_path(path) == so.json_hash_path(path)
    fri = so.Fri(so.GENERATOR, so.primitive_nth_root(256), 256, 4, 17)
    for seed in (b"", b"seed", bytes(range(32))):
        assert D.fri_sample_indices(seed, 128, 8, 6) == fri.sample_indices(seed, 128, 8, 6)
    for n, e, t in ((1 << 24, 4, 17), (256, 4, 17), (512, 4, 2), (64, 8, 2)):
        assert D.fri_num_rounds(n, e, t) == so.Fri(so.GENERATOR, so.primitive_nth_root(n), n, e, t).num_rounds()
    assert D.fri_num_rounds(1 << 24, 4, 17) == 18 and D.fri_num_rounds(4, 4, 1) == 0


def test_shard_layout_and_period():
    from stark_amd import distributed as D
    n, world, period = 64, 4, 16
    full = torch.arange(2 * n, dtype=torch.int64).view(n, 2)
    seen = []
    for r in range(world):
        sh = D.fri_shard_of(full, r, world, period)
        pos = (sh[:, 0] // 2).tolist()
        c = period // world
        assert pos == [q * period + r * c + j for q in range(n // period) for j in range(c)]
        # a position and its fold partner are in the same share
        assert all(((p + n // 2) % n) in pos for p in pos)
        seen += pos
    assert sorted(seen) == list(range(n))
    # the period is raised so that the last round is never a sharded one, and capped at n
    assert D.fri_shard_period(1 << 24, 8, 4, 17) == 1 << 17
    assert D.fri_shard_period(1 << 10, 4, 4, 2, 4) == 16
    assert D.fri_shard_period(1 << 9, 2, 4, 2, None) == 1 << 9
    with pytest.raises(ValueError, match="less than 1 round"):
        D.fri_shard_period(4, 2, 4, 1)

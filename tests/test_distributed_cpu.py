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

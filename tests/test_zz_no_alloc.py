"""The entry point that allocates nothing (stark_merkle_root_fe_static_dev, a static device array as
its tree scratch) against stark_merkle_commit_fe_dev, through the C ABI: bit-exact roots on two calls
in a row, the second with a smaller input, plus the size limit and the edge sizes."""
import ctypes as C

import numpy as np
import pytest

import oracle_c as oc
from stark_amd import Context, StarkError

pytestmark = pytest.mark.gpu


class _Dev:
    """n bytes of device memory through the C ABI's own allocator."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.p = ctx, C.c_void_p()
        ctx._chk(ctx.lib.stark_dev_alloc(ctx.h, nbytes, C.byref(self.p)))

    def data_ptr(self):
        return self.p.value

    def free(self):
        self.ctx._chk(self.ctx.lib.stark_dev_free(self.ctx.h, self.p))


def _upload(ctx, a):
    d = _Dev(ctx, a.nbytes)
    ctx._chk(ctx.lib.stark_copy_h2d(ctx.h, d.p, a.ctypes.data, a.nbytes))
    return d


def _static_root(ctx, elems_dev, n, root_dev):
    ctx.merkle_root_fe_static_dev(elems_dev, n, root_dev)
    out = np.zeros(32, dtype=np.uint8)
    ctx._chk(ctx.lib.stark_copy_d2h(ctx.h, out.ctypes.data, root_dev.p, 32))     # ordered behind the kernels, synchronises
    return out.tobytes()


def test_static_scratch_root_matches_allocating_entry_point():
    ctx = Context(0)
    try:
        big = oc.splitmix(1 << 18, 11)            # the maximum
        small = oc.splitmix(1 << 9, 12)           # a smaller input right after: the array still holds the big tree
        d_big, d_small = _upload(ctx, big), _upload(ctx, small)
        root_dev = _Dev(ctx, 32)
        r1 = _static_root(ctx, d_big, 1 << 18, root_dev)
        r2 = _static_root(ctx, d_small, 1 << 9, root_dev)
        assert r1 == ctx.merkle_commit_fe_dev(d_big, 1 << 18)
        assert r2 == ctx.merkle_commit_fe_dev(d_small, 1 << 9)
        assert r2 == oc.merkle_commit_fe(small)
        # edge sizes: a single leaf is its own root; two leaves; structured values with short leaf encodings
        edge = np.zeros((4, 2), dtype=np.uint64)
        edge[1, 0] = 1
        edge[2, 0] = 1 << 40
        edge[3] = oc.fe(407 << 119)                  # p - 1
        d_edge = _upload(ctx, edge)
        for n in (1, 2, 4):
            assert _static_root(ctx, d_edge, n, root_dev) == ctx.merkle_commit_fe_dev(d_edge, n)
        # again after the small ones: nothing stale is read
        assert _static_root(ctx, d_big, 1 << 18, root_dev) == r1
        # beyond the maximum: refused with a clear error, nothing launched
        with pytest.raises(StarkError) as e:
            ctx.merkle_root_fe_static_dev(d_big, 1 << 19, root_dev)
        assert e.value.code == 2 and "static scratch" in e.value.message
        with pytest.raises(StarkError):
            ctx.merkle_root_fe_static_dev(d_big, 3, root_dev)
        for d in (d_big, d_small, d_edge, root_dev):
            d.free()
    finally:
        ctx.close()

"""Host-pointer transfers taken apart: stark_copy_h2d and stark_copy_d2h on pageable numpy memory, per size.
usage: python3 tools/edge_split_time.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from stark_amd import Context
from stark_amd.synth import field_vector

ctx = Context(0)
lib = ctx.lib
for log_n in (19, 20, 21, 22, 23, 24):
    n = 1 << log_n
    x = field_vector(n, log_n)
    y = np.empty_like(x)
    d = C.c_void_p()
    ctx._chk(lib.stark_dev_alloc(ctx.h, C.c_size_t(16 * n), C.byref(d)))
    res = {}
    for name, fn in (("h2d", lambda: lib.stark_copy_h2d(ctx.h, d, C.c_void_p(x.ctypes.data), C.c_size_t(16 * n))),
                     ("d2h", lambda: lib.stark_copy_d2h(ctx.h, C.c_void_p(y.ctypes.data), d, C.c_size_t(16 * n)))):
        ctx._chk(fn())
        ctx._chk(lib.stark_ctx_synchronize(ctx.h))
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            ctx._chk(fn())
            ctx._chk(lib.stark_ctx_synchronize(ctx.h))
            ts.append(time.perf_counter() - t0)
        res[name] = sorted(ts)[len(ts) // 2]
    assert np.array_equal(x, y)
    ctx._chk(lib.stark_dev_free(ctx.h, d))
    print("2^%d (%4d MiB): h2d %.2f ms = %.1f GB/s, d2h %.2f ms = %.1f GB/s"
          % (log_n, 16 * n >> 20, res["h2d"] * 1e3, 16 * n / res["h2d"] / 1e9, res["d2h"] * 1e3, 16 * n / res["d2h"] / 1e9))
ctx.close()

"""Wall-clock of the host-pointer entry points (pageable numpy memory in and out, PCIe included).
usage: python3 tools/host_api_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from stark_amd import Context
from stark_amd.synth import field_vector

ctx = Context(0)
for log_n in (20, 22, 24):
    n = 1 << log_n
    x = field_vector(n, log_n)
    buf = x.copy()                      # transformed in place, as NTT::forward does (ntt.rs:25-28)

    def median_wall(fn, reps=7):        # single calls now and then take several times as long (host page placement)
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    dt = median_wall(lambda: ctx._chk(ctx.lib.stark_ntt(ctx.h, buf.ctypes.data, n, 0)))
    dm = median_wall(lambda: ctx.merkle_commit_fe(x))
    print("2^%d: stark_ntt %.2f ms (%.1f GB/s over both directions), stark_merkle_commit_fe %.2f ms (%.1f GB/s in)"
          % (log_n, dt * 1e3, 32 * n / dt / 1e9, dm * 1e3, 16 * n / dm / 1e9))
ctx.close()

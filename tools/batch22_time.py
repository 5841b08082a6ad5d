"""2^22-point forward transforms alone and in batches (columns side by side): what a launch that fills the
chip evenly gives per transform, against the single transform's 1024 tiles on 768 workgroup slots.
usage: python3 tools/batch22_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stark_amd import Context
from stark_amd.synth import field_vector

torch.cuda.set_device(0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for log_n in (20, 21, 22):
    n = 1 << log_n
    for cols in (1, 2, 3, 4, 6, 12):
        d = torch.from_numpy(field_vector(n * cols, 7).view(np.int64)).cuda()
        fn = lambda: ctx.ntt_dev(d, n, inverse=False, n_cols=cols, col_stride=n)  # noqa: E731
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print("2^%d x %2d columns: %.4f ms, %.4f ms per transform, %.1f Gelem/s" % (log_n, cols, ms, ms / cols, n * cols / ms / 1e6), flush=True)
ctx.close()

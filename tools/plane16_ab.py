"""The pass before the last reading its twiddles from the 20-byte limb plane against the 16-byte canonical plane
(STARK_NTT_PLANE16_FROM=<log n>), alternating between two contexts of one process, results compared bit for bit.
usage: python3 tools/plane16_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stark_amd import Context
from stark_amd.synth import field_vector

torch.cuda.set_device(0)
os.environ.pop("STARK_NTT_PLANE16_FROM", None)
a = Context(0)
os.environ["STARK_NTT_PLANE16_FROM"] = "12"
b = Context(0)
for c in (a, b):
    c.set_stream(torch.cuda.current_stream().cuda_stream)
for log_n in (16, 20, 22, 24, 26):
    n = 1 << log_n
    x = torch.from_numpy(field_vector(n, 7).view(np.int64)).cuda()
    ya, yb = x.clone(), x.clone()
    a.ntt_dev(ya, n, False)
    b.ntt_dev(yb, n, False)
    torch.cuda.synchronize()
    same = bool(torch.equal(ya, yb))
    a.ntt_dev(ya, n, True)
    b.ntt_dev(yb, n, True)
    torch.cuda.synchronize()
    same = same and bool(torch.equal(ya, x)) and bool(torch.equal(yb, x))
    res = {0: [], 1: []}
    for rnd in range(8):
        for which in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            c, d = (a, ya) if which == 0 else (b, yb)
            for _ in range(3):
                c.ntt_dev(d, n, False)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                c.ntt_dev(d, n, False)
            e1.record()
            e1.synchronize()
            res[which].append(e0.elapsed_time(e1) / 10)
    print("2^%d same=%s  20-byte plane %s | 16-byte plane %s" % (
        log_n, same, " ".join("%.4f" % v for v in sorted(res[0])[:4]), " ".join("%.4f" % v for v in sorted(res[1])[:4])),
        flush=True)

"""Batched small transforms (the local steps of the four-step transform): 2^24 elements as
cols x n for several n.  usage: python3 tools/small_batch_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_amd import Context
from stark_amd.synth import field_vector

ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
total = 1 << 24
d = torch.from_numpy(field_vector(total, 5).view(np.int64)).cuda()
for log_n in (4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 24):
    n = 1 << log_n
    cols = total // n
    for _ in range(2):
        ctx.ntt_dev(d, n, False, n_cols=cols, col_stride=n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctx.ntt_dev(d, n, False, n_cols=cols, col_stride=n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("n=2^%-2d cols=%-6d %.3f ms  %.1f Gelem/s  (%.2f ps per element-stage)" % (log_n, cols, ms, total / ms / 1e6, ms * 1e9 / total / log_n))
ctx.close()

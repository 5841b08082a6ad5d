"""Commitments of several 2^24-element codewords with their trees retained (as the prover makes them), timed one by one.
usage: python3 tools/merkle_keep_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_amd import Context
from stark_amd.synth import field_vector

ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << 24
bufs = [torch.from_numpy(field_vector(n, 5 + i).view(np.int64)).cuda() for i in range(3)]
for rnd in range(4):
    trees, times = [], []
    for b in bufs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        root, tree = ctx.merkle_commit_fe_dev(b, n, keep_tree=True)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        trees.append(tree)
    t0 = time.perf_counter()
    for t in trees:
        t.free()
    free_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    r = ctx.merkle_commit_fe_dev(bufs[0], n)
    torch.cuda.synchronize()
    print("round", rnd, "kept trees:", " ".join("%.2f" % t for t in times), "ms; frees %.2f ms; without tree %.2f ms" %
          (free_ms, (time.perf_counter() - t0) * 1e3))
ctx.close()

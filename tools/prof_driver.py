"""Small driver for profiling one workload at a time under rocprofv3.
usage: python3 tools/prof_driver.py {ntt24|ntt22|merkle24|fri24|lde} [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_amd import Context, GENERATOR, hostlib
from stark_amd.synth import field_vector

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
if what in ("ntt24", "ntt22"):
    n = 1 << int(what[3:])
    d = torch.from_numpy(field_vector(n, 1).view(np.int64)).to(dev)
    for _ in range(reps):
        ctx.ntt_dev(d, n, False)
        ctx.ntt_dev(d, n, True)
elif what == "merkle24":
    n = 1 << 24
    d = torch.from_numpy(field_vector(n, 1).view(np.int64)).to(dev)
    for _ in range(reps):
        ctx.merkle_commit_fe_dev(d, n)
elif what == "fri24":
    n = 1 << 24
    d = torch.from_numpy(field_vector(n, 1).view(np.int64)).to(dev)
    omega = hostlib.primitive_nth_root(n)
    for _ in range(reps):
        hostlib.fri_commit_resident(d.data_ptr(), n, GENERATOR, omega, 4, 17, hostlib.Stream())
elif what == "lde":
    cols = 16
    c = torch.from_numpy(field_vector(cols << 20, 3).view(np.int64)).to(dev)
    out = torch.empty((cols << 22, 2), dtype=torch.int64, device=dev)
    for _ in range(reps):
        ctx.coset_lde_batch_dev(c, 1 << 20, 1 << 20, cols, 1 << 22, GENERATOR, out, 1 << 22)
torch.cuda.synchronize()
ctx.close()
print("done", what)

"""Wall-clock of stark_fri_commit_dev at 2^24 with a Python transcript (STARK_HIP_LIBRARY selects the build).
usage: python3 tools/fri_time.py [reps]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_amd import Context, GENERATOR
from stark_amd.synth import field_vector

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P = (407 << 119) + 1
n = 1 << 24
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d = torch.from_numpy(field_vector(n, 24).view(np.int64)).cuda()
omega = pow(GENERATOR, (1 << 119) >> 24, P)


def once():
    objs = []

    def transcript(rnd, last, root):
        objs.append(root.hex())
        if last:
            return None
        ser = len(objs).to_bytes(8, "little") + b"".join(len(o).to_bytes(8, "little") + o.encode() for o in objs)
        return int.from_bytes(hashlib.shake_256(ser).digest(32), "big") % P
    ctx.fri_commit(d, GENERATOR, omega, 4, 17, transcript, n=n).free()


once()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    once()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("STARK_HIP_LIBRARY", "in-tree"), "fri commit 2^24: min %.3f median %.3f ms" % (min(ts), sorted(ts)[len(ts) // 2]))
ctx.close()

"""One-rank run of the sharded FRI commit at 2^24 (for kernel traces and wall-clock timing).
usage: python3 tools/fri_sharded_time.py [reps]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_amd import Context, GENERATOR
from stark_amd.distributed import HipEngine, fri_commit_sharded
from stark_amd.synth import field_vector

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = (407 << 119) + 1
n = 1 << 24
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
eng = HipEngine(ctx)
share = torch.from_numpy(field_vector(n, 3000).view(np.int64)).cuda()
omega = pow(GENERATOR, (1 << 119) >> 24, P)


def once():
    objs = []

    def transcript(rnd, last, root):
        objs.append(root.hex())
        if last:
            return None
        ser = len(objs).to_bytes(8, "little") + b"".join(len(o).to_bytes(8, "little") + o.encode() for o in objs)
        return int.from_bytes(hashlib.shake_256(ser).digest(32), "big") % P
    fri_commit_sharded(eng, share, n, GENERATOR, omega, 4, 17, transcript).free()


once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    once()
torch.cuda.synchronize()
print("sharded commit 2^24, one rank: %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3))
ctx.close()

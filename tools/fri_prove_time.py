"""Wall-clock of Fri::prove (commit + query phase) at 2^24 through the C++ mirror, resident codeword.
usage: python3 tools/fri_prove_time.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stark_amd import GENERATOR, hostlib
from stark_amd.synth import field_vector

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 1 << 24
torch.cuda.set_device(0)
# a codeword the verifier accepts: 2^22 random coefficients extended to the coset of 2^24 points
from stark_amd import Context
ctx = Context(0)
coeffs = torch.from_numpy(field_vector(n // 4, 24).view(np.int64)).cuda()
d = torch.empty((n, 2), dtype=torch.int64, device="cuda")
ctx.coset_lde_batch_dev(coeffs, n // 4, n // 4, 1, n, GENERATOR, d, n)
ctx.synchronize()
omega = hostlib.primitive_nth_root(n)


def commit_only():
    hostlib.fri_commit_resident(d.data_ptr(), n, GENERATOR, omega, 4, 17, hostlib.Stream())


def prove():
    ps = hostlib.Stream()
    assert len(hostlib.fri_prove_resident(d.data_ptr(), n, GENERATOR, omega, 4, 17, ps)) == 34
    return ps


for name, fn in (("commit", commit_only), ("prove", prove)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("Fri::%s 2^24: %.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))
ps = prove()
print("proof stream:", len(ps.objects), "objects,", len(ps.serialize()), "bytes; verifier accepts:",
      hostlib.fri_verify(GENERATOR, omega, n, 4, 17, hostlib.Stream.deserialize(ps.serialize()))[0])

"""Forward against inverse transform time, alternating, with per-kernel times from HIP events.
usage: python3 tools/ntt_fwd_inv_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stark_amd import Context
from stark_amd.synth import field_vector

torch.cuda.set_device(0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for log_n in (22, 24):
    n = 1 << log_n
    d = torch.from_numpy(field_vector(n, 7).view(np.int64)).cuda()
    for inv in (False, True):
        for _ in range(3):
            ctx.ntt_dev(d, n, inv)
    res = {False: [], True: []}
    for rnd in range(6):
        for inv in ((False, True) if rnd % 2 == 0 else (True, False)):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ctx.ntt_dev(d, n, inv)
            e1.record(); e1.synchronize()
            res[inv].append(e0.elapsed_time(e1) / 10)
    print("2^%d forward %s" % (log_n, " ".join("%.4f" % v for v in res[False])))
    print("2^%d inverse %s" % (log_n, " ".join("%.4f" % v for v in res[True])))

"""Time forward NTTs of 2^22 and 2^24 points on device-resident data (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stark_amd import Context
from stark_amd.synth import field_vector

torch.cuda.set_device(0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
out = []
for log_n in (22, 24):
    n = 1 << log_n
    d = torch.from_numpy(field_vector(n, 7).view(np.int64)).cuda()
    for _ in range(5):
        ctx.ntt_dev(d, n, False)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ctx.ntt_dev(d, n, False)
    e1.record(); e1.synchronize()
    out.append(f"2^{log_n}: {e0.elapsed_time(e1) / reps:.4f} ms")
print(os.environ.get("STARK_HIP_LIBRARY", "default"), " ".join(out))

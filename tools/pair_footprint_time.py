"""Is the forward+inverse pair at 2^22 slower than its halves because its working set (data, ping buffer and
TWO 84 MB twiddle planes = 302 MB) no longer fits the 256 MB last-level cache?  Times, per transform: forward
only; forward + inverse on one buffer; forward alternating between two buffers (one plane, 285 MB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stark_amd import Context
from stark_amd.synth import field_vector


def timed(fn, reps=40):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
import statistics, time
for log_n in (21, 22):
    n = 1 << log_n
    A = torch.from_numpy(field_vector(n, 1).view(np.int64)).cuda()
    B = torch.from_numpy(field_vector(n, 2).view(np.int64)).cuda()
    cases = {
        "fwd": (lambda: ctx.ntt_dev(A, n, False), 1),
        "inv": (lambda: ctx.ntt_dev(A, n, True), 1),
        "pair/2": (lambda: (ctx.ntt_dev(A, n, False), ctx.ntt_dev(A, n, True)), 2),
        "two buffers/2": (lambda: (ctx.ntt_dev(A, n, False), ctx.ntt_dev(B, n, False)), 2),
    }
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:                     # clock ramp
        cases["pair/2"][0]()
    torch.cuda.synchronize()
    res = {k: [] for k in cases}
    for rnd in range(7):
        for k, (fn, div) in cases.items():
            res[k].append(timed(fn) / div)
    print(f"2^{log_n}: " + "  ".join(f"{k} {statistics.median(v)*1e3:.1f} us" for k, v in res.items()), flush=True)
ctx.close()

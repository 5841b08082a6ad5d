import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import oracle_c as oc, stark_oracle as so
from stark_amd import Context
ctx = Context(0)
for log_n in (25, 26):
    n = 1 << log_n
    x = oc.splitmix(n, log_n)
    t0 = time.time(); ref = oc.ntt(x); t1 = time.time()
    f = ctx.ntt(x)
    print(log_n, 'forward equal', np.array_equal(f, ref), 'oracle s', round(t1 - t0, 1), flush=True)
    b = ctx.ntt(f, inverse=True)
    print(log_n, 'roundtrip', np.array_equal(b, x), flush=True)
    del ref, f, b
for log_n in (27, 28, 30):
    n = 1 << log_n
    x = oc.splitmix(n, log_n)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    t0 = time.time()
    ctx.ntt_dev(d, n, False); torch.cuda.synchronize()
    print(log_n, 'fwd ms', (time.time() - t0) * 1e3, flush=True)
    f = d.cpu().numpy().view(np.uint64)
    # spot check: X[k] = sum x_j w^(jk) for a few k, by Horner over the whole vector in the C oracle? use python ints on a strided subset is not exact; use DFT property instead:
    # X[0] = sum x_j ; X[n/2] = sum (-1)^j x_j
    P = so.P
    s_all = 0; s_alt = 0
    lo = x[:, 0].astype(object); hi = x[:, 1].astype(object)
    # sums via numpy on 32-bit pieces to stay exact
    def total(a):
        return int(a.astype(np.uint64).__and__(np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64)) + (int((a >> np.uint64(32)).sum(dtype=np.uint64)) << 32)
    se_lo, se_hi = total(x[0::2, 0]), total(x[0::2, 1])
    so_lo, so_hi = total(x[1::2, 0]), total(x[1::2, 1])
    ev = (se_lo + (se_hi << 64)) % P; od = (so_lo + (so_hi << 64)) % P
    print(log_n, 'X[0]', oc.to_int(f[0]) == (ev + od) % P, 'X[n/2]', oc.to_int(f[n // 2]) == (ev - od) % P, flush=True)
    ctx.ntt_dev(d, n, True); torch.cuda.synchronize()
    print(log_n, 'roundtrip', np.array_equal(d.cpu().numpy().view(np.uint64), x), flush=True)
    del d, f
ctx.close()

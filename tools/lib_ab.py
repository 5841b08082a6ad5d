"""Several builds of libstark_hip.so on the same box, alternating: each measurement is a fresh process
(STARK_HIP_LIBRARY selects the build) that ramps the clocks for 300 ms and then times the 2^22 forward+inverse
pair, the 2^24 forward transform and the 16-column 2^20 -> 2^22 extension with HIP events.
usage: python3 tools/lib_ab.py [rounds] name=path ...        (path "-" = the in-tree build)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from stark_amd import Context, GENERATOR
    from stark_amd.synth import field_vector
    torch.cuda.set_device(0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def ev(fn, reps):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    n22, n24 = 1 << 22, 1 << 24
    d22 = torch.from_numpy(field_vector(n22, 7).view(np.int64)).cuda()

    def pair():
        ctx.ntt_dev(d22, n22, False)
        ctx.ntt_dev(d22, n22, True)
    a = ev(pair, 200)
    d24 = torch.from_numpy(field_vector(n24, 8).view(np.int64)).cuda()
    b = ev(lambda: ctx.ntt_dev(d24, n24, False), 50)
    c = ev(lambda: ctx.ntt_dev(d24, n24, True), 50)
    del d24
    cols = 16
    coeffs = torch.from_numpy(field_vector(cols << 20, 3).view(np.int64)).cuda()
    out = torch.empty((cols << 22, 2), dtype=torch.int64, device="cuda")
    e = ev(lambda: ctx.coset_lde_batch_dev(coeffs, 1 << 20, 1 << 20, cols, 1 << 22, GENERATOR, out, 1 << 22), 10)
    print("%.4f %.4f %.4f %.4f" % (a, b, c, e))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    args = sys.argv[1:]
    rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
    builds = [a.split("=", 1) for a in args] or [["in-tree", "-"]]
    res = {name: [] for name, _ in builds}
    for _ in range(rounds):
        for name, path in builds:
            env = dict(os.environ)
            env.pop("STARK_HIP_LIBRARY", None)
            if path != "-":
                env["STARK_HIP_LIBRARY"] = os.path.abspath(path)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True,
                                 text=True, timeout=120)
            if out.returncode != 0:
                sys.exit("%s failed:\n%s" % (name, out.stderr[-2000:]))
            res[name].append([float(v) for v in out.stdout.split()])
    print("%-10s %-26s %-26s %-26s %-26s" % ("build", "2^22 fwd+inv ms", "2^24 fwd ms", "2^24 inv ms", "16 x 2^20->2^22 ms"))
    for name, rows in res.items():
        print("%-10s " % name + " ".join("%-26s" % " ".join("%.4f" % r[k] for r in rows) for k in range(4)))

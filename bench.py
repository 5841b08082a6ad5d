#!/usr/bin/env python3
"""Benchmark of the STARK prover hot path on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on
rank 0.  A step is one forward + one inverse 2^22-point NTT over the 128-bit
field on device-resident data; with N > 1 every rank
transforms its own vector (independent units, no collective: weak scaling) and
`value` is the whole-job rate.  The same line carries the roofline of the
dominant kernel family and the CPU restatement timed on the host cores, plus
the other headline numbers (2^24 NTT, 2^24 FRI commit, batched coset LDE) under
"extra".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_N = 22
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
# From the rocprofv3 PMC passes kept under profiles/ (r01_ntt22_pmc_{fetch,write,valu}.csv; see
# profiles/r01_summary.txt), means over the three pass kernels of a 2^22 transform:
# HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 counts half of a wide read stream) + WRITE_SIZE
PMC_TRAFFIC_BYTES_PER_LAUNCH = (2 * 47915 + 65536) * 1024     # 28 MB of it is the middle pass's twiddle plane
# 64-bit-class VALU wave-instructions per launch (v_mad_i64_i32 and the 64-bit shifts/adds of the
# Montgomery product), and all VALU wave-instructions per launch
PMC_INT64_WAVE_INSTR_PER_LAUNCH = 9661099
PMC_VALU_WAVE_INSTR_PER_LAUNCH = 17555456
# a SIMD issues one 64-bit-class integer instruction per 4 cycles: 1024 SIMDs x 2.4 GHz / 4
VALU_INT64_PEAK_GINSTR_S = 1024 * 2.4 / 4


def _ramp(torch, fn, ms=120.0):
    """Untimed load: repeats fn() for about `ms` so that the measurement that follows sees the
    sustained clocks (the GPU falls back to its idle state within milliseconds without work).
    Only for calls without collectives: the repeat count differs from rank to rank."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        fn()
        torch.cuda.synchronize()


def _events_ms(torch, fn, reps):
    """Average milliseconds of fn() over reps, HIP events on the current stream."""
    _ramp(torch, fn)
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def cpu_baseline(log_n: int, reps: int):
    """The C restatement (oracle/stark_oracle.c, one thread) on the same
    workload: forward + inverse NTT of 2^log_n elements."""
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):   # the checker lives outside the package
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_c as oc
    n = 1 << log_n
    x = oc.splitmix(n, 0x5741524B)
    t0 = time.perf_counter()
    for _ in range(reps):
        y = oc.ntt(x)
        oc.ntt(y, inverse=True)
    dt = time.perf_counter() - t0
    return {
        "value": round(2 * n * reps / dt / 1e6, 4), "unit": "Melem/s", "cores": 1, "kind": "port",
        "sample": f"{reps} x (forward+inverse) NTT at 2^{log_n}, single thread, C restatement of ntt.rs",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="untimed load before the warm-up steps so that the GPU has left its idle power state")
    ap.add_argument("--no-extra", action="store_true", help="skip the 2^24 / FRI / LDE side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    from stark_amd import Context, GENERATOR
    from stark_amd.synth import field_vector

    ctx = Context(dev.index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    n = 1 << LOG_N
    host = field_vector(n, 0x5741524B + rank)
    data = torch.from_numpy(host.view(np.int64)).to(dev)
    orig = data.clone()

    def step():
        ctx.ntt_dev(data, n, inverse=False)
        ctx.ntt_dev(data, n, inverse=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # An idle MI355X takes some tens of milliseconds of load to reach its sustained clocks (the same
    # step measures 0.34 ms right after start-up and 0.285 ms from then on); the ramp is untimed and
    # comes before the W warm-up steps, which are done as asked.
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_stop = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_start.record()
    for _ in range(args.steps):
        step()
    ev_stop.record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms_total = ev_start.elapsed_time(ev_stop)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the transform pair must give the input back, bit for bit
    assert torch.equal(data, orig), "forward+inverse NTT did not round-trip"

    result = None
    if rank == 0:
        passes_per_ntt = 3                                     # 2^22 = 2^8 * 2^7 * 2^7
        launches = args.steps * 2 * passes_per_ntt
        avg_launch_ms = kernel_ms_total / launches
        alg_bytes = 32.0 * n                                   # each pass reads and writes every element once
        achieved = alg_bytes / (avg_launch_ms * 1e-3) / 1e9
        result = {
            "metric": "128-bit-field NTT throughput, 2^22-point forward+inverse",
            "value": round(world * 2 * n * args.steps / dt / 1e6, 2),
            "unit": "Melem/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u128 (mod 407*2^119+1)", "data": "synthetic",
            "config": {"workload": "single 2^22-point forward+inverse NTT per GPU, data resident in HBM",
                       "log_n": LOG_N, "elements_per_step": 2 * n, "parallelism": f"independent x{world}",
                       "ramp_ms": args.ramp_ms},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH,
                         "algorithmic_bytes": int(alg_bytes),
                         "kernel": "ntt_pass_kernel (3 launches per transform)",
                         "avg_launch_us": round(avg_launch_ms * 1e3, 2),
                         # the butterfly passes are integer-multiply bound: 64-bit multiply-add issue
                         # rate (counter SQ_INSTS_VALU_INT64) against one such instruction per SIMD per 4 cycles
                         "valu_int64": {"achieved": round(PMC_INT64_WAVE_INSTR_PER_LAUNCH / (avg_launch_ms * 1e-3) / 1e9, 1),
                                        "peak": VALU_INT64_PEAK_GINSTR_S, "unit": "Gwave-instr/s",
                                        "frac": round(PMC_INT64_WAVE_INSTR_PER_LAUNCH / (avg_launch_ms * 1e-3) / 1e9
                                                      / VALU_INT64_PEAK_GINSTR_S, 4),
                                        "all_valu_wave_instr_per_launch": PMC_VALU_WAVE_INSTR_PER_LAUNCH}},
        }

    extra = {}
    if not args.no_extra and rank == 0:
        # forward-only and inverse-only at 2^22
        extra["ntt_2^22_fwd_ms"] = round(_events_ms(torch, lambda: ctx.ntt_dev(data, n, False), 20), 4)
        extra["ntt_2^22_inv_ms"] = round(_events_ms(torch, lambda: ctx.ntt_dev(data, n, True), 20), 4)
        extra["ntt_2^22_fwd_Melem_s"] = round(n / extra["ntt_2^22_fwd_ms"] / 1e3, 1)
        # 2^24 forward / inverse
        n24 = 1 << 24
        d24 = torch.from_numpy(field_vector(n24, 24).view(np.int64)).to(dev)
        for _ in range(3):
            ctx.ntt_dev(d24, n24, False)
        extra["ntt_2^24_fwd_ms"] = round(_events_ms(torch, lambda: ctx.ntt_dev(d24, n24, False), 10), 4)
        extra["ntt_2^24_inv_ms"] = round(_events_ms(torch, lambda: ctx.ntt_dev(d24, n24, True), 10), 4)
        extra["ntt_2^24_fwd_Melem_s"] = round(n24 / extra["ntt_2^24_fwd_ms"] / 1e3, 1)
        # Merkle commitment of 2^24 leaves
        _ramp(torch, lambda: ctx.merkle_commit_fe_dev(d24, n24))
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.merkle_commit_fe_dev(d24, n24)
        torch.cuda.synchronize()
        extra["merkle_2^24_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        extra["merkle_2^24_Mhash_s"] = round((2 * n24 - 1) / extra["merkle_2^24_ms"] / 1e3, 1)
        # FRI commit on a 2^24 codeword (blowup 4, 17 colinearity tests: 18 roots, 17 folds) through
        # the C++ host mirror of Fri::commit: the Fiat-Shamir transcript is its native ProofStream
        from stark_amd import hostlib
        omega = hostlib.primitive_nth_root(n24)

        def fri_once():
            ps = hostlib.Stream()
            hostlib.fri_commit_resident(d24.data_ptr(), n24, GENERATOR, omega, 4, 17, ps)

        _ramp(torch, fri_once)
        t0 = time.perf_counter()
        for _ in range(3):
            fri_once()
        torch.cuda.synchronize()
        extra["fri_commit_2^24_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)

        # Fri::prove = the commit plus the query phase answered from the resident layers (one round
        # trip per round); the codeword is a real one (2^22 coefficients extended to 2^24 points)
        cf = torch.from_numpy(field_vector(n24 // 4, 25).view(np.int64)).to(dev)
        cw24 = torch.empty((n24, 2), dtype=torch.int64, device=dev)
        ctx.coset_lde_batch_dev(cf, n24 // 4, n24 // 4, 1, n24, GENERATOR, cw24, n24)
        torch.cuda.synchronize()
        _ramp(torch, lambda: hostlib.fri_prove_resident(cw24.data_ptr(), n24, GENERATOR, omega, 4, 17, hostlib.Stream()))
        t0 = time.perf_counter()
        for _ in range(3):
            hostlib.fri_prove_resident(cw24.data_ptr(), n24, GENERATOR, omega, 4, 17, hostlib.Stream())
        torch.cuda.synchronize()
        extra["fri_prove_2^24_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        del cf, cw24

        # the same commit with the transcript in a Python callback over the C ABI
        import hashlib

        def fri_py_once():
            objs = []

            def transcript(rnd, last, root):
                objs.append(root.hex())
                if last:
                    return None
                ser = len(objs).to_bytes(8, "little") + b"".join(len(o).to_bytes(8, "little") + o.encode() for o in objs)
                return int.from_bytes(hashlib.shake_256(ser).digest(32), "big") % ((407 << 119) + 1)
            ctx.fri_commit(d24, GENERATOR, omega, 4, 17, transcript, n=n24).free()

        _ramp(torch, fri_py_once)
        t0 = time.perf_counter()
        for _ in range(3):
            fri_py_once()
        torch.cuda.synchronize()
        extra["fri_commit_2^24_python_transcript_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        del d24
        # batched coset LDE: 16 of the 64 columns of config 3 (2^20 coefficients -> 2^22 points each)
        cols = 16
        coeffs = torch.from_numpy(field_vector(cols << 20, 3).view(np.int64)).to(dev)
        out = torch.empty((cols << 22, 2), dtype=torch.int64, device=dev)
        lde = lambda: ctx.coset_lde_batch_dev(coeffs, 1 << 20, 1 << 20, cols, 1 << 22, GENERATOR, out, 1 << 22)  # noqa: E731
        lde()
        extra["coset_lde_16x2^20_to_2^22_ms"] = round(_events_ms(torch, lde, 5), 3)
        extra["coset_lde_Melem_s_out"] = round((cols << 22) / extra["coset_lde_16x2^20_to_2^22_ms"] / 1e3, 1)
        del coeffs, out

    # side measurements every rank takes part in: the batched-LDE workload (64 columns of 2^20
    # coefficients -> 2^22 points each, one Merkle root per column), sharded column-wise with no
    # data-path collective (only the 32-byte roots are gathered); and, with more than one GPU, the
    # four-step 2^24 transform whose transposes are RCCL all-to-alls
    multi = {}
    if not args.no_extra:
        try:
            from stark_amd.distributed import HipEngine, fourstep_ntt, lde_commit_columns, shard_columns
            eng = HipEngine(ctx)
            mine = shard_columns(64, rank, world)
            cl = torch.from_numpy(field_vector(len(mine) << 20, 2000 + rank).view(np.int64)).to(dev)
            cl = cl.view(len(mine), 1 << 20, 2)
            for _ in range(3):                                   # same count on every rank (root gather inside)
                lde_commit_columns(eng, cl, 64, 1 << 22, GENERATOR)
            barrier()
            t0 = time.perf_counter()
            lde_commit_columns(eng, cl, 64, 1 << 22, GENERATOR)
            barrier()
            ms = (time.perf_counter() - t0) * 1e3
            multi["lde_commit_64x2^20_to_2^22_ms"] = round(ms, 3)
            multi["lde_commit_Melem_s_out"] = round((64 << 22) / ms / 1e3, 1)
            del cl
            # FRI commit of one 2^24 codeword dealt out in chunks over the ranks (folds and hashing
            # local, 32-byte subtree roots all-gathered per round, the small layers on every rank)
            if world & (world - 1) == 0:
                import hashlib
                from stark_amd.distributed import fri_commit_sharded
                n24 = 1 << 24
                w24 = pow(GENERATOR, (1 << 119) >> 24, (407 << 119) + 1)      # primitive_nth_root(2^24)
                share = torch.from_numpy(field_vector(n24 // world, 3000 + rank).view(np.int64)).to(dev)

                def fri_sharded_once():
                    objs = []

                    def transcript(rnd, last, root):
                        objs.append(root.hex())
                        if last:
                            return None
                        ser = len(objs).to_bytes(8, "little") + b"".join(
                            len(o).to_bytes(8, "little") + o.encode() for o in objs)
                        return int.from_bytes(hashlib.shake_256(ser).digest(32), "big") % ((407 << 119) + 1)
                    fri_commit_sharded(eng, share, n24, GENERATOR, w24, 4, 17, transcript).free()

                for _ in range(15):                              # same count on every rank (collectives inside)
                    fri_sharded_once()
                barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    fri_sharded_once()
                barrier()
                multi["fri_commit_2^24_sharded_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
                del share
            if world > 1:
                n24 = 1 << 24
                per = n24 // world
                xl = torch.from_numpy(field_vector(per, 1000 + rank).view(np.int64)).to(dev)
                for _ in range(40):
                    fourstep_ntt(xl, n24, eng)
                barrier()
                t0 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    fourstep_ntt(xl, n24, eng)
                barrier()
                ms = (time.perf_counter() - t0) / reps * 1e3
                multi["fourstep_ntt_2^24_ms"] = round(ms, 4)
                multi["fourstep_ntt_2^24_Melem_s"] = round(n24 / ms / 1e3, 1)
                barrier()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fourstep_ntt(xl, n24, eng, natural_output=False)
                barrier()
                multi["fourstep_ntt_2^24_transposed_out_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
                del xl
        except Exception as e:  # noqa: BLE001 - side measurements must not lose the headline line
            multi["error"] = repr(e)

    if rank == 0:
        extra.update(multi)
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(LOG_N, 6)      # about 12 s of CPU work
        else:
            result["cpu_baseline"] = None
        result["extra"] = extra
        print(json.dumps(result), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

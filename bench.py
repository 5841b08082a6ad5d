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
N_SIMD = 256 * 4               # CUs x SIMDs per CU
NOMINAL_GHZ = 2.4              # peak engine clock
# Every VALU instruction of these kernels holds its SIMD for one quad-cycle (SQ_ACTIVE_INST_VALU equals
# SQ_INSTS_VALU in the profiles), so the wave-instruction rate has the hardware ceiling SIMDs x clock / 4.
VALU_PEAK_GINSTR_S = N_SIMD * NOMINAL_GHZ / 4
def _latest_pmc_file():
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r03_pmc.json")


PMC_FILE = _latest_pmc_file()    # the newest round's counter profile


def load_pmc():
    """Counter figures of the committed profile (written by tools/make_profile_summary.py from
    rocprofv3 --pmc passes).  They describe the kernels as they were when the profile was taken: the
    file carries a hash of the kernel sources, and when that no longer matches the sources next to
    this script the figures are not quoted (traffic = null)."""
    from stark_amd._lib import kernel_source_stamp
    try:
        with open(PMC_FILE) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None, "no counter profile committed"
    if pmc.get("kernel_source_sha256") != kernel_source_stamp():
        return None, "stale: the kernel sources changed after %s was taken" % os.path.relpath(PMC_FILE, ROOT)
    return pmc, os.path.relpath(PMC_FILE, ROOT)


def valu_roofline(valu_wave_instr: float, seconds: float, clock_ghz_profiled=None):
    """VALU-issue roofline: wave-instructions per second (SQ_INSTS_VALU of the profiled kernels over the time
    measured here) against 1024 SIMDs x 2.4 GHz / 4, and against the same ceiling at the clock the chip held
    while the counters were taken (SQ_BUSY_CYCLES / 32 over the dispatches' wall time)."""
    rate = valu_wave_instr / seconds / 1e9
    r = {"bound": "valu", "achieved": round(rate, 1), "peak": round(VALU_PEAK_GINSTR_S, 1), "unit": "Gwave-instr/s",
         "frac": round(rate / VALU_PEAK_GINSTR_S, 4), "peak_clock_ghz": NOMINAL_GHZ,
         "valu_wave_instr": int(valu_wave_instr)}
    if clock_ghz_profiled:
        peak_p = N_SIMD * clock_ghz_profiled / 4.0
        r.update({"clock_ghz_profiled": clock_ghz_profiled, "peak_at_profiled_clock": round(peak_p, 1),
                  "frac_at_profiled_clock": round(min(rate / peak_p, 1.0), 4)})
        if rate > peak_p:                                   # this run held a higher clock than the profiled one did
            r["clock_above_profiled"] = True
    return r


def ntt_roofline(pmc, source, key, n, avg_launch_ms, kernel):
    """Roofline of one pass launch of a 2^k transform.  The passes are bound by VALU issue (the modular product is
    about fifty instructions and the SIMDs are busy 75-90 % of the kernels' cycles), so that is the primary figure
    when the committed counters describe the kernels as built; the HBM figure (32 bytes per element and pass
    against 8 TB/s) is kept beside it under "hbm".  Without fresh counters only the HBM figure can be given."""
    alg = 32.0 * n                                             # a pass reads and writes every element once
    seconds = avg_launch_ms * 1e-3
    hbm = {"bound": "hbm", "achieved": round(alg / seconds / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(alg / seconds / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(alg)}
    common = {"traffic": None, "kernel": kernel, "avg_launch_us": round(avg_launch_ms * 1e3, 2), "counters": source}
    if not (pmc and key in pmc):
        return {**hbm, **common, "note": "VALU-issue bound kernel; its instruction counts come from the counter profile, "
                                         "which does not match these sources, so only the HBM figure is given"}
    e = pmc[key]
    vi = e.get("valu_issue") or {}
    r = valu_roofline(e["valu_wave_instr_per_launch"], seconds, vi.get("clock_ghz"))
    r.update(common)
    r["traffic"] = int(e["traffic_bytes_per_launch"])
    hbm["traffic_over_algorithmic"] = round(e["traffic_bytes_per_launch"] / alg, 3)
    r["hbm"] = hbm
    i64 = e["int64_wave_instr_per_launch"]
    # a lower estimate of the issue time: the 64-bit-class instructions (multiply-adds, 64-bit shifts) at their measured
    # 4.2 cycles and the rest at the 2.4 cycles a two-source 32-bit instruction costs when several waves share the SIMD
    # (tools/ubench*.hip); the primary figure prices every instruction at 4
    if vi.get("clock_ghz"):
        cyc = (4.2 * i64 + 2.4 * (e["valu_wave_instr_per_launch"] - i64)) / N_SIMD
        r["frac_weighted_issue_cost_at_profiled_clock"] = round(cyc / (seconds * vi["clock_ghz"] * 1e9), 4)
    r["int64_class_wave_instr"] = int(i64)
    r["busy_frac_profiled"] = vi.get("busy_frac")
    r["waves_per_simd_profiled"] = vi.get("waves_per_simd")
    # SQ_WAIT_INST_ANY / SQ_WAIT_ANY against SQ_WAVE_CYCLES: with about three waves per SIMD, a wave that issues a
    # third of the time means a saturated issue port
    r["wave_cycles_split"] = e.get("wave_cycles_split")
    return r


def hash_roofline(pmc, source, key, compressions, seconds, algorithmic_bytes, extra_fields=None):
    """Merkle / FRI commitment: BLAKE2b on 32-bit VALU lanes is issue bound.  Primary figure as valu_roofline; the
    compression rate and the compute-only ceiling of the same hash code (tools/hash_bench.hip, a measurement, not a
    hardware limit) are secondary fields."""
    gh = compressions / seconds / 1e9
    e = (pmc or {}).get(key)
    if e:
        r = valu_roofline(e["valu_wave_instr_per_call"], seconds, (e.get("valu_issue") or {}).get("clock_ghz"))
        r["traffic"] = int(e["traffic_bytes_per_call"])
        r["launches_per_call"] = e.get("launches_per_call")
    else:
        r = {"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_GINSTR_S, 1), "unit": "Gwave-instr/s",
             "frac": None, "traffic": None,
             "note": "instruction counts come from the counter profile, which does not match these sources"}
    ceil = (pmc or {}).get("hash_ceiling_ghash_s", {})
    r.update({"compressions": compressions, "Gcompression_s": round(gh, 2),
              "hash_bench_ceiling_Gcompression_s": min(ceil.values()) if ceil else None,
              "hash_bench_note": "compute-only rate of the same hash code measured by tools/hash_bench.hip on the "
                                 "profiled box; a self-measured secondary figure, not the roofline's peak",
              "algorithmic_bytes": int(algorithmic_bytes),
              "hbm_GB_s": round(algorithmic_bytes / seconds / 1e9, 1), "counters": source})
    if extra_fields:
        r.update(extra_fields)
    return r


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


def _oracle():
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):   # the checker lives outside the package
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_c as oc
    return oc


def host_threads() -> int:
    """Threads for the all-cores CPU lines: this process's share of the host, not the machine's core count."""
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        if os.environ.get(var, "").isdigit():
            share = min(share, int(os.environ[var]))
    return max(1, min(share, 16))


def cpu_baseline(log_n: int, reps: int):
    """The C restatement (oracle/stark_oracle.c, one thread as the reference is single-threaded) on the same
    workload: forward + inverse NTT of 2^log_n elements."""
    oc = _oracle()
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


def cpu_baseline_all_cores(log_n: int, reps: int):
    """The same restatement with its butterfly loops shared by all the host threads this process may use."""
    oc = _oracle()
    n, k = 1 << log_n, host_threads()
    x = oc.splitmix(n, 0x5741524B)
    with oc.threads(k):
        oc.ntt(x)                                              # starts the thread pool
        t0 = time.perf_counter()
        for _ in range(reps):
            y = oc.ntt(x)
            oc.ntt(y, inverse=True)
        dt = time.perf_counter() - t0
    return {
        "value": round(2 * n * reps / dt / 1e6, 4), "unit": "Melem/s", "cores": k, "kind": "port",
        "sample": f"{reps} x (forward+inverse) NTT at 2^{log_n}, {k} threads (OpenMP over the butterflies of each stage), "
                  "C restatement of ntt.rs; not the reference, which is single-threaded",
    }


def fri_commit_cpu():
    """Fri::commit (fri.rs:100-152) of the C restatement: one thread at 2^20 (the 2^24 figure is that times 16,
    the work being linear in the length), and all host threads at the full 2^24."""
    oc = _oracle()
    import stark_oracle as so
    out = {"kind": "port", "unit": "ms", "params": "blowup 4, 17 colinearity tests"}
    n20 = 1 << 20
    x = oc.splitmix(n20, 0x4652)
    t0 = time.perf_counter()
    oc.fri_commit(x, so.GENERATOR, so.primitive_nth_root(n20), 4, 17)
    ms = (time.perf_counter() - t0) * 1e3
    out["one_thread_2^20_ms"] = round(ms, 1)
    out["one_thread_2^24_ms_extrapolated"] = round(16 * ms, 1)
    k = host_threads()
    n24 = 1 << 24
    x = oc.splitmix(n24, 0x4653)
    with oc.threads(k):
        t0 = time.perf_counter()
        oc.fri_commit(x, so.GENERATOR, so.primitive_nth_root(n24), 4, 17)
        out["all_threads_2^24_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["cores"] = k
    return out


def api_edge(ctx, field_vector):
    """The host-pointer entry points as a binding that holds a Vec<FieldElement> meets them (ntt.rs:25-40 takes the
    vector by reference and transforms it in place; Merkle::commit takes the leaves): pageable host memory in and
    out, PCIe included.  Outside the timed region; never the headline value."""
    out = {"note": "wall time of stark_ntt / stark_merkle_commit_fe on pageable host memory (copy in, kernels, copy out): "
                   "median of 7 calls after one untimed call; single calls now and then take two to four times as long "
                   "(page placement on the host side), which the worst of the seven shows"}

    def median_wall(fn, reps=7):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], max(ts)

    for log_n in (22, 24):
        n = 1 << log_n
        buf = field_vector(n, log_n)
        dt, worst = median_wall(lambda: ctx._chk(ctx.lib.stark_ntt(ctx.h, buf.ctypes.data, n, 0)))
        out["stark_ntt_2^%d_ms" % log_n] = round(dt * 1e3, 3)
        out["stark_ntt_2^%d_worst_ms" % log_n] = round(worst * 1e3, 3)
        out["stark_ntt_2^%d_Melem_s" % log_n] = round(n / dt / 1e6, 1)
        out["stark_ntt_2^%d_pcie_GB_s_both_ways" % log_n] = round(32 * n / dt / 1e9, 1)
        if log_n == 24:
            dm, _ = median_wall(lambda: ctx.merkle_commit_fe(buf))
            out["stark_merkle_commit_fe_2^24_ms"] = round(dm * 1e3, 3)
            out["stark_merkle_commit_fe_2^24_pcie_GB_s_in"] = round(16 * n / dm / 1e9, 1)
    return out


def launch_ranks(n_ranks: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N fresh processes of this script, one per GPU,
    with the rendezvous variables torch.distributed reads, relay rank 0's JSON line and return the worst exit code.
    This parent never touches the GPU (it imports neither torch nor the library), so nothing that has initialised
    HIP is ever replaced or forked."""
    import socket
    import subprocess
    with socket.socket() as s:                                   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n_ranks), RANK=str(r),
                   LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n_ranks))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    worst, failed_at = 0, None
    live = set(range(n_ranks))
    while live:
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or rc
                failed_at = failed_at or time.monotonic()
        # a rank that died leaves the others waiting in a collective: give them a moment, then end them
        if failed_at is not None and live and time.monotonic() - failed_at > 20:
            for r in live:
                procs[r].kill()
        time.sleep(0.05)
    reader.join(timeout=10)
    for l in lines:                                              # the line, not what a transport library printed beside it
        (sys.stdout if l.lstrip().startswith("{") else sys.stderr).write(l)
    sys.stdout.flush()
    return worst


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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # under a launcher (torchrun) WORLD_SIZE is what counts; --gpus is then only a label

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal of the N > 1 code path where only one GPU exists: every rank uses GPU 0 and gloo carries the
    # collectives (STARK_BENCH_SHARE_GPU=1).  Its numbers mean nothing; it is there to run the code.
    share_gpu = world > 1 and os.environ.get("STARK_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        os.environ["STARK_DEVICE"] = "0"                         # the C++ mirror's context follows LOCAL_RANK otherwise
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 and not share_gpu else 0)

    from stark_amd import Context, GENERATOR
    from stark_amd.synth import field_vector

    # what the collective layer saw: backend, world size and the device every rank runs on
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": dev.index, "name": props.name, "gcn_arch": getattr(props, "gcnArchName", None)}
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        collective = {"backend": dist.get_backend(), "world": dist.get_world_size(), "devices": everyone}
    else:
        collective = {"backend": None, "world": 1, "devices": [mine]}

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
        t = torch.tensor([dt], device="cpu" if share_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the transform pair must give the input back, bit for bit
    assert torch.equal(data, orig), "forward+inverse NTT did not round-trip"

    result = None
    pmc, pmc_source = load_pmc() if rank == 0 else (None, None)
    if rank == 0:
        passes_per_ntt = 3                                     # 2^22 = 2^8 * 2^7 * 2^7
        launches = args.steps * 2 * passes_per_ntt
        avg_launch_ms = kernel_ms_total / launches
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
                       "ramp_ms": args.ramp_ms, **({"rehearsal": "all ranks on one GPU, gloo"} if share_gpu else {})},
            "roofline": ntt_roofline(pmc, pmc_source, "ntt22", n, avg_launch_ms,
                                     "ntt_pass_kernel (3 launches per transform)"),
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
        extra["ntt_2^24_roofline"] = ntt_roofline(pmc, pmc_source, "ntt24", n24, extra["ntt_2^24_fwd_ms"] / 3,
                                                  "ntt_pass_kernel (3 launches per forward transform, 2^8 x 2^8 x 2^8)")
        # Merkle commitment of 2^24 leaves
        _ramp(torch, lambda: ctx.merkle_commit_fe_dev(d24, n24))
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.merkle_commit_fe_dev(d24, n24)
        torch.cuda.synchronize()
        extra["merkle_2^24_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        extra["merkle_2^24_Mhash_s"] = round((2 * n24 - 1) / extra["merkle_2^24_ms"] / 1e3, 1)
        # 96 N bytes: leaves and the two levels above them in one pass (16 N read, 56 N written), the levels above
        # through HBM (16 N read, 8 N written); 144 N with one launch per level, 80 N if the whole retained tree
        # were written in one pass
        extra["merkle_roofline"] = hash_roofline(pmc, pmc_source, "merkle24", 2 * n24 - 1, extra["merkle_2^24_ms"] * 1e-3,
                                                 96 * n24)
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
        # 18 layers of 2^24 .. 2^7 elements: one tree each (2 n_r - 1 compressions) and 17 folds of 32 n_r bytes;
        # every root goes to the host transcript and comes back as the next fold's challenge
        fri_comp = sum(2 * (n24 >> r) - 1 for r in range(18))
        fri_fold_bytes = sum(32 * (n24 >> r) for r in range(17))
        fri_tree_bytes = sum((96 if (n24 >> r) > (1 << 17) else 144) * (n24 >> r) for r in range(18))
        extra["fri_commit_roofline"] = hash_roofline(
            pmc, pmc_source, "fri24", fri_comp, extra["fri_commit_2^24_ms"] * 1e-3, fri_fold_bytes + fri_tree_bytes,
            {"fold_bytes": fri_fold_bytes, "tree_bytes": fri_tree_bytes, "host_round_trips": 17,
             "round_trip_note": "per round: root written to host-mapped memory by the kernel that computes it, stream "
                                "synchronised, hex + SHAKE256 + sample on the host, next fold launched"})

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
        # the scaled workload: Stark::prove of a 2^20-row Rescue-Prime chain (2 registers, FRI domain 2^24,
        # blowup 4, 2 colinearity tests) with the polynomial algebra on the device; the trace itself (the
        # computation being proven) is made once on the host and is not part of the proof time
        try:
            import ctypes as C
            import hashlib as _hl
            hostlib.lib.sh_chain_new.restype = C.c_void_p
            hostlib.lib.sh_chain_random_bytes.restype = C.c_size_t
            ctx.trim()                                            # the prover allocates through its own context
            t0 = time.perf_counter()
            job = C.c_void_p(hostlib.lib.sh_chain_new(hostlib.fe(2024), C.c_size_t(1 << 20), C.c_size_t(4), C.c_size_t(2),
                                                      C.c_size_t(2)))
            if not job:
                raise RuntimeError(hostlib.lib.sh_last_panic().decode())
            extra["stark_trace_and_tables_2^20_s"] = round(time.perf_counter() - t0, 2)
            rb = _hl.shake_256(b"bench").digest(hostlib.lib.sh_chain_random_bytes(job))
            rbuf = (C.c_uint8 * len(rb)).from_buffer_copy(rb)
            proof = (C.c_uint8 * (1 << 20))()
            names = ("host", "interpolate", "quotients", "lde", "merkle", "fri", "openings", "total")
            def proofs(fn, count):
                out = []
                for _ in range(count):
                    plen.value = len(proof)
                    tm = (C.c_double * 8)()
                    hostlib.chk(fn(job, rbuf, C.c_size_t(len(rb)), proof, C.byref(plen), tm))
                    out.append(list(tm))
                return sorted(out[1:], key=lambda t: t[7])        # the first proof builds tables: not counted

            plen = C.c_size_t(len(proof))
            runs = proofs(hostlib.lib.sh_chain_prove_columns, 9)  # the trace handed over by columns
            best, median = runs[0], runs[len(runs) // 2]
            rows_runs = proofs(hostlib.lib.sh_chain_prove, 5)     # row by row, as Stark::prove takes it (stark.rs:223-228)
            out_el = (C.c_uint64 * 2)()
            hostlib.chk(hostlib.lib.sh_chain_output(job, out_el))
            t0 = time.perf_counter()
            accepted = hostlib.lib.sh_chain_verify(job, out_el, proof, plen)
            extra["stark_verify_2^20_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            assert accepted == 1, "the verifier rejected the 2^20-row proof"
            extra["stark_prove_2^20_ms"] = round(median[7], 2)
            extra["stark_prove_2^20_note"] = ("median of 8 proofs (after one untimed proof) through ScaledStark::prove_columns, "
                                              "the trace handed over column by column; best under stark_prove_2^20_best_ms, "
                                              "the row-by-row entry (host transposition included) under stark_prove_2^20_rows_entry_ms")
            extra["stark_prove_2^20_rows_entry_ms"] = round(rows_runs[len(rows_runs) // 2][7], 2)
            extra["stark_prove_2^20_best_ms"] = round(best[7], 2)
            extra["stark_prove_2^20_split_ms"] = {k: round(v, 2) for k, v in zip(names, median)}
            extra["stark_proof_2^20_bytes"] = plen.value
            extra["stark_proof_2^20_sha256"] = _hl.sha256(bytes(proof[:plen.value])).hexdigest()
            hostlib.lib.sh_chain_free(job)
        except Exception as e:  # noqa: BLE001 - a side measurement must not lose the headline line
            extra["stark_prove_2^20_error"] = repr(e)

    # side measurements every rank takes part in: the batched-LDE workload (64 columns of 2^20
    # coefficients -> 2^22 points each, one Merkle root per column), sharded column-wise with no
    # data-path collective (only the 32-byte roots are gathered); and, with more than one GPU, the
    # four-step 2^24 transform whose transposes are RCCL all-to-alls
    multi = {}
    # A collective that never completes (a peer that died, an exchange the fabric does not finish) must not cost
    # the headline line: past the deadline rank 0 prints what it has, and every rank leaves without waiting for the
    # others.  Exit code 0 everywhere, so that a launcher does not count the lost side measurement as a failed run.
    watchdog = None
    if world > 1 and not args.no_extra:
        import threading
        deadline_s = float(os.environ.get("STARK_BENCH_MULTI_DEADLINE_S", "240"))
        # rank 0 arrives here after its single-GPU side measurements, the others wait for it in the first exchange
        patience_s = deadline_s if rank == 0 else deadline_s + 180.0

        def expire():
            if rank == 0:
                late = dict(extra)
                late.update(multi)
                late["collective"] = collective
                late["multi_gpu_error"] = ("the measurements that need every rank did not finish within %.0f s; "
                                           "the line carries what was measured until then" % deadline_s)
                result["cpu_baseline"] = None
                result["extra"] = late
                print(json.dumps(result), flush=True)
            os._exit(0)

        watchdog = threading.Timer(patience_s, expire)
        watchdog.daemon = True
        watchdog.start()
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
            if world > 1 and world & (world - 1) == 0:
                # the 2^20-row proof with everything on the FRI domain (extensions, commitments, FRI, openings)
                # spread over the ranks; the steps off that domain run on every rank alike
                import hashlib as _hl2
                from stark_amd import hostlib as _hostlib
                from stark_amd import scaled, sharded_stark
                ctx.trim()
                mctx = scaled.mirror_context()
                meng = HipEngine(mctx)
                job = scaled.ChainJob(2024, 1 << 20, 4, 2, 2)
                rb = _hl2.shake_256(b"bench").digest(job.random_bytes)
                runs = []
                for i in range(7):
                    barrier()
                    t0 = time.perf_counter()
                    steps = scaled.ChainSteps(job, rb, mctx)
                    tm = {}
                    proof = sharded_stark.stark_prove_sharded(meng, steps, _hostlib.Stream(), timings=tm)
                    barrier()
                    ms = (time.perf_counter() - t0) * 1e3
                    if i >= 2:
                        runs.append((ms, dict(tm), dict(steps.timings)))
                    steps.free()
                runs.sort(key=lambda r: r[0])
                ms, tm, st = runs[len(runs) // 2]
                multi["stark_prove_2^20_sharded_ms"] = round(ms, 2)
                multi["stark_prove_2^20_sharded_best_ms"] = round(runs[0][0], 2)
                multi["stark_prove_2^20_sharded_split_ms"] = {
                    "replicated_steps": {k: round(st[k], 2) for k in ("host", "interpolate", "quotients")},
                    "sharded": {k: round(tm[k], 2) for k in ("lde", "merkle", "fri", "openings")},
                    "note": "median of 5 (rank 0's laps); replicated steps run on every rank alike"}
                multi["stark_prove_2^20_sharded_proof_sha256"] = _hl2.sha256(proof).hexdigest()
                job.free()
                del meng, mctx
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
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0:
        extra.update(multi)
        extra["collective"] = collective
        if not args.no_extra:
            try:
                extra["api_edge"] = api_edge(ctx, field_vector)
            except Exception as e:  # noqa: BLE001 - a side measurement must not lose the headline line
                extra["api_edge"] = {"error": repr(e)}
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(LOG_N, 3)              # about 6 s of CPU work on one thread
            result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(LOG_N, 3)
            try:
                extra["fri_commit_cpu"] = fri_commit_cpu()
            except Exception as e:  # noqa: BLE001
                extra["fri_commit_cpu"] = {"error": repr(e)}
        else:
            result["cpu_baseline"] = None
        result["extra"] = extra
        print(json.dumps(result), flush=True)

    ctx.close()
    if world > 1:
        try:                                                   # the line is out: a peer that has already gone is no failure of this run
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


if __name__ == "__main__":
    main()

"""Turns the raw output of tools/profile_round.sh into the files kept under profiles/:
<round>_pmc.json (the counter figures bench.py quotes, stamped with the kernel sources they were taken
for), <round>_summary.txt, and copies of the CSVs they come from.

usage: python3 tools/make_profile_summary.py <round, e.g. r02> <raw dir, e.g. gpurun_out/prof_r02>"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stark_amd._lib import kernel_source_stamp  # noqa: E402

rnd, raw = sys.argv[1], os.path.abspath(sys.argv[2])
P = os.path.join(ROOT, "profiles")
WORKLOADS = {"ntt22": 5, "ntt24": 5, "merkle24": 3, "fri24": 3}      # repetitions prof_driver.py ran (profile_round.sh)
SETUP_KERNELS = ("ntt_plane_kernel", "rocclr", "fillBuffer", "copyBuffer")


def find(dirname, suffix):
    for base, _, files in os.walk(os.path.join(raw, dirname)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(base, f)
    raise FileNotFoundError(f"{dirname}/*{suffix}")


def short(name):
    name = name.replace("void ", "").replace("starkk::", "")
    return name[:name.index("(")] if "(" in name else name


def pmc(workload, group):
    """{kernel: {counter: [value per dispatch]}} of one counter pass, set-up kernels dropped."""
    path = find(f"{workload}_{group}", "counter_collection.csv")
    shutil.copy(path, os.path.join(P, f"{rnd}_{workload}_pmc_{group}.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if any(s in k for s in SETUP_KERNELS):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def pmc_dispatches(workload, group):
    """[{kernel, counters..., dur_ns}] per dispatch of one counter pass (set-up kernels dropped)."""
    path = find(f"{workload}_{group}", "counter_collection.csv")
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if any(s in k for s in SETUP_KERNELS):
            continue
        e = d.setdefault(r["Dispatch_Id"], {"kernel": k, "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(d.values())


N_SIMD = 256 * 4          # CUs x SIMDs
N_SE = 32                 # shader engines (8 XCDs x 4): SQ_BUSY_CYCLES is summed over them
N_XCD = 8                 # GRBM_GUI_ACTIVE is summed over the XCDs


def kernel_stats(workload):
    path = find(f"{workload}_kt", "kernel_stats.csv")
    shutil.copy(path, os.path.join(P, f"{rnd}_{workload}_kernel_stats.csv"))
    return {short(r["Name"]): (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]))
            for r in csv.DictReader(open(path))}


out = []
w = out.append
data = {"kernel_source_sha256": kernel_source_stamp(),
        "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip(),
        "note": "FETCH_SIZE on gfx950 reports half of a wide coalesced read stream: reads = 2 x FETCH_SIZE (KB); "
                "WRITE_SIZE (KB) is exact for 16-byte-per-lane stores. Per-launch / per-call means over the profiled "
                "repetitions; each counter group was collected in a rocprofv3 run of its own."}
w(f"Profile summary {rnd} (MI355X, gfx950, ROCm 7.2) -- written by tools/make_profile_summary.py from the output of")
w("tools/profile_round.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20, then per workload")
w("(tools/prof_driver.py ntt22|ntt24|merkle24|fri24) one rocprofv3 --pmc run per counter group and one kernel trace.")
w(f"kernel sources sha256 {data['kernel_source_sha256'][:16]}..., commit {data['commit'][:10]}")
w("")
bench_log = open(os.path.join(raw, "bench.log")).read().strip().splitlines()
line = json.loads([l for l in bench_log if l.startswith("{")][-1])
open(os.path.join(P, f"{rnd}_bench_line.json"), "w").write(json.dumps(line) + "\n")
w("bench line of the profiled run: value %.1f %s, %.4f ms/step, roofline avg_launch_us %.2f" %
  (line["value"], line["unit"], line["ms_per_step"], line["roofline"]["avg_launch_us"]))
w("")
shutil.copy(find("bench", "kernel_stats.csv"), os.path.join(P, f"{rnd}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(raw, "bench_ntt_launches_by_grid.csv"), os.path.join(P, f"{rnd}_bench_ntt_launches_by_grid.csv"))
w("%-92s %6s %14s %12s %6s" % ("kernel (bench run)", "calls", "total_ns", "avg_ns", "pct"))
for r in csv.DictReader(open(os.path.join(P, f"{rnd}_bench_kernel_stats.csv"))):
    if float(r["Percentage"]) >= 0.5:
        w("%-92s %6s %14s %12.0f %6.2f" % (short(r["Name"])[:92], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                          float(r["Percentage"])))
w("")
w("ntt_pass_kernel launches of the bench run split by size:")
for r in csv.DictReader(open(os.path.join(P, f"{rnd}_bench_ntt_launches_by_grid.csv"))):
    w("  %-40s %9s elements %6s calls  avg %8s ns  min %8s ns" % (r["kernel"], r["elements"], r["calls"], r["avg_ns"], r["min_ns"]))

for wl, reps in WORKLOADS.items():
    f, wr, v, kt = pmc(wl, "fetch"), pmc(wl, "write"), pmc(wl, "valu"), kernel_stats(wl)
    w("")
    w(f"== {wl}: {reps} repetitions of tools/prof_driver.py {wl}")
    entry = {"kernels": {}}
    tot_read = tot_write = tot_i64 = tot_valu = 0.0
    launches = 0
    for k in sorted(f):
        fs, ws = f[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
        vi, v64 = v[k]["SQ_INSTS_VALU"], v[k]["SQ_INSTS_VALU_INT64"]
        calls, _, avg_ns = kt.get(k, (len(fs), 0.0, 0.0))
        entry["kernels"][k] = {"launches": len(fs), "fetch_kb_mean": sum(fs) / len(fs), "write_kb_mean": sum(ws) / len(ws),
                               "valu_wave_instr_mean": sum(vi) / len(vi), "int64_wave_instr_mean": sum(v64) / len(v64),
                               "avg_ns": avg_ns}
        w("  %-44s %4d launches  FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  SQ_INSTS_VALU %12.0f  INT64 %12.0f  avg %9.0f ns" %
          (k[:44], len(fs), sum(fs) / len(fs), sum(ws) / len(ws), sum(vi) / len(vi), sum(v64) / len(v64), avg_ns))
        tot_read += 2 * sum(fs) * 1024
        tot_write += sum(ws) * 1024
        tot_i64 += sum(v64)
        tot_valu += sum(vi)
        launches += len(fs)
    if wl.startswith("ntt"):
        passes = [k for k in f if "ntt_pass_kernel" in k]
        n_l = sum(len(f[k]["FETCH_SIZE"]) for k in passes)
        tr = sum(2 * sum(f[k]["FETCH_SIZE"]) + sum(wr[k]["WRITE_SIZE"]) for k in passes) * 1024 / n_l
        entry["traffic_bytes_per_launch"] = tr
        entry["int64_wave_instr_per_launch"] = sum(sum(v[k]["SQ_INSTS_VALU_INT64"]) for k in passes) / n_l
        entry["valu_wave_instr_per_launch"] = sum(sum(v[k]["SQ_INSTS_VALU"]) for k in passes) / n_l
        entry["pass_launches_profiled"] = n_l
        # where a resident wave's cycles go (guide: WAIT_ANY = parked on s_waitcnt or a barrier, WAIT_INST_ANY =
        # ready but not issued because the pipe is taken, the rest = issuing; the three are disjoint)
        wc = sum(sum(v[k]["SQ_WAVE_CYCLES"]) for k in passes)
        wa = sum(sum(v[k]["SQ_WAIT_ANY"]) for k in passes)
        wi = sum(sum(v[k]["SQ_WAIT_INST_ANY"]) for k in passes)
        entry["wave_cycles_split"] = {"issuing": round(1 - (wa + wi) / wc, 4), "issue_stall": round(wi / wc, 4),
                                      "waitcnt_or_barrier": round(wa / wc, 4)}
        # VALU occupation of the SIMDs.  SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles; it equals
        # SQ_INSTS_VALU here, i.e. every VALU instruction of these kernels holds its SIMD for one quad-cycle),
        # SQ_BUSY_CYCLES counts shader cycles per shader engine, so busy/32 is the kernel's length in cycles and
        # that over the dispatch's wall time is the clock the chip held (GRBM_GUI_ACTIVE / 8 cross-checks it).
        disp = [x for x in pmc_dispatches(wl, "valu") if "ntt_pass_kernel" in x["kernel"]]
        cyc = sum(x["SQ_BUSY_CYCLES"] for x in disp) / N_SE
        entry["valu_issue"] = {
            "busy_frac": round(4 * sum(x["SQ_ACTIVE_INST_VALU"] for x in disp) / (N_SIMD * cyc), 4),
            "waves_per_simd": round(4 * sum(x["SQ_WAVE_CYCLES"] for x in disp) / (N_SIMD * cyc), 3),
            "clock_ghz": round(cyc / sum(x["dur_ns"] for x in disp), 3),
            "cycles_per_launch": round(cyc / len(disp)),
            "per_kernel": {}}
        for k in passes:
            dk = [x for x in disp if x["kernel"] == k]
            ck = sum(x["SQ_BUSY_CYCLES"] for x in dk) / N_SE
            entry["valu_issue"]["per_kernel"][k] = {
                "busy_frac": round(4 * sum(x["SQ_ACTIVE_INST_VALU"] for x in dk) / (N_SIMD * ck), 4),
                "waves_per_simd": round(4 * sum(x["SQ_WAVE_CYCLES"] for x in dk) / (N_SIMD * ck), 3),
                "valu_instr_per_wave": round(sum(x["SQ_INSTS_VALU"] for x in dk) / len(dk) / ((1 << int(wl[3:])) / 1024)),
                "cycles": round(ck / len(dk))}
        try:
            g = [x for x in pmc_dispatches(wl, "grbm") if "ntt_pass_kernel" in x["kernel"]]
            shutil.copy(find(f"{wl}_grbm", "counter_collection.csv"), os.path.join(P, f"{rnd}_{wl}_pmc_grbm.csv"))
            entry["valu_issue"]["clock_ghz_grbm"] = round(sum(x["GRBM_GUI_ACTIVE"] for x in g) / N_XCD / sum(x["dur_ns"] for x in g), 3)
        except FileNotFoundError:
            pass
        n = 1 << int(wl[3:])
        w("  mean over the %d pass launches: HBM-side traffic %.1f MB per launch (algorithmic 32 B x 2^%s = %.1f MB), "
          "%.0f 64-bit-class of %.0f VALU wave-instructions" % (n_l, tr / 1e6, wl[3:], 32 * n / 1e6,
                                                                 entry["int64_wave_instr_per_launch"], entry["valu_wave_instr_per_launch"]))
        w("  a resident wave's cycles: %.0f %% issuing, %.0f %% ready but held back by a busy pipe (SQ_WAIT_INST_ANY), "
          "%.0f %% parked on s_waitcnt or a barrier (SQ_WAIT_ANY)" % tuple(100 * entry["wave_cycles_split"][k] for k in
                                                                            ("issuing", "issue_stall", "waitcnt_or_barrier")))
        vi_ = entry["valu_issue"]
        w("  VALU pipes busy %.0f %% of the kernels' cycles (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32)), "
          "%.2f waves resident per SIMD on average, clock held %.2f GHz (SQ_BUSY_CYCLES / 32 / wall%s)" %
          (100 * vi_["busy_frac"], vi_["waves_per_simd"], vi_["clock_ghz"],
           "; GRBM_GUI_ACTIVE / 8 / wall = %.2f" % vi_["clock_ghz_grbm"] if "clock_ghz_grbm" in vi_ else ""))
        for k, e in vi_["per_kernel"].items():
            w("    %-44s busy %.0f %%  %.2f waves/SIMD  %d VALU instructions per wave  %d cycles" %
              (k[:44], 100 * e["busy_frac"], e["waves_per_simd"], e["valu_instr_per_wave"], e["cycles"]))
    else:
        entry["traffic_bytes_per_call"] = (tot_read + tot_write) / reps
        entry["valu_wave_instr_per_call"] = tot_valu / reps
        entry["launches_per_call"] = launches / reps
        w("  per call: HBM-side traffic %.1f MB in %.1f launches, %.0f VALU wave-instructions" %
          (entry["traffic_bytes_per_call"] / 1e6, entry["launches_per_call"], entry["valu_wave_instr_per_call"]))
        # the clock the chip held over the call's kernels and how busy the VALU pipes were (as for the transforms)
        disp = pmc_dispatches(wl, "valu")
        cyc = sum(x["SQ_BUSY_CYCLES"] for x in disp) / N_SE
        dur = sum(x["dur_ns"] for x in disp)
        entry["valu_issue"] = {
            "busy_frac": round(4 * sum(x["SQ_ACTIVE_INST_VALU"] for x in disp) / (N_SIMD * cyc), 4),
            "clock_ghz": round(cyc / dur, 3),
            "kernel_ms_per_call": round(dur / reps / 1e6, 4),
            "valu_ginstr_s_in_kernels": round(tot_valu / dur, 1)}
        w("  VALU pipes busy %.0f %% of the kernels' cycles, clock held %.2f GHz, %.3f ms of kernel time per call: "
          "%.0f Gwave-instr/s inside the kernels against 1024 x %.2f / 4 = %.0f at that clock and 614 at 2.4 GHz" %
          (100 * entry["valu_issue"]["busy_frac"], entry["valu_issue"]["clock_ghz"], entry["valu_issue"]["kernel_ms_per_call"],
           entry["valu_issue"]["valu_ginstr_s_in_kernels"], entry["valu_issue"]["clock_ghz"],
           N_SIMD * entry["valu_issue"]["clock_ghz"] / 4))
    data[wl] = entry

hb = os.path.join(raw, "hash_bench.txt")
if os.path.exists(hb):
    shutil.copy(hb, os.path.join(P, f"{rnd}_hash_bench.txt"))
    w("")
    w("== tools/hash_bench.hip (compute-only ceiling of the BLAKE2b compressions, chained in registers)")
    ceil = {}
    for l in open(hb):
        w("  " + l.rstrip())
        if "Ghash/s" in l and "compute only" in l:
            ceil[l.split("(")[0].strip()] = float(l.split("Ghash/s")[0].split()[-1])
    data["hash_ceiling_ghash_s"] = ceil
with open(os.path.join(P, f"{rnd}_pmc.json"), "w") as fjson:
    json.dump(data, fjson, indent=1, sort_keys=True)
open(os.path.join(P, f"{rnd}_summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))

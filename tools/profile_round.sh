#!/bin/bash
# Collects the profiles kept under profiles/: kernel stats of the bench command and the PMC
# passes of the 2^22 transform (each counter group in a run of its own).
# usage (on the GPU box, from the repository root): bash tools/profile_round.sh <out_dir>
set -e
R=$PWD
OUT=$R/${1:-gpurun_out/prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 20 > $OUT/bench.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 $R/tools/prof_driver.py ntt22 5 > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 $R/tools/prof_driver.py ntt22 5 > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/valu -o valu --output-format csv -- python3 $R/tools/prof_driver.py ntt22 5 > $OUT/valu.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/valu24 -o valu24 --output-format csv -- python3 $R/tools/prof_driver.py ntt24 5 > $OUT/valu24.log 2>&1
python3 - "$OUT" <<'PY'
# ntt_pass_kernel launches of the bench run split by grid size (copied to profiles/<round>_bench_ntt_launches_by_grid.csv)
import collections, csv, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(out + "/bench/bench_kernel_trace.csv")):
    k = r["Kernel_Name"]
    if "ntt_pass_kernel" in k:
        agg[(k[k.index("ntt_pass_kernel"):k.index("(")], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/bench_ntt_launches_by_grid.csv", "w") as f:
    f.write("kernel,grid_threads,elements,calls,avg_ns,min_ns\n")
    for (name, g), v in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        f.write('"%s",%d,%d,%d,%.0f,%d\n' % (name, g, g * 16, len(v), sum(v) / len(v), min(v)))
PY
ls $OUT/*

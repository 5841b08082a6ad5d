#!/bin/bash
# Collects what profiles/<round>_* is made from: kernel stats of the bench command, the PMC passes of
# the 2^22 and 2^24 transforms, of the 2^24 Merkle commitment and of the 2^24 FRI commit (each counter
# group in a run of its own; for the transforms also GRBM_GUI_ACTIVE, the clock cross-check), and the
# compute-only hash ceiling of tools/hash_bench.hip.
# usage (on the GPU box, from the repository root): bash tools/profile_round.sh <out_dir>
set -e
R=$PWD
OUT=$R/${1:-gpurun_out/prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 20 > $OUT/bench.log 2>&1
VALU="SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for W in ntt22 ntt24 merkle24 fri24; do
    REPS=5; if [ $W = merkle24 ] || [ $W = fri24 ]; then REPS=3; fi
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/${W}_fetch -o fetch --output-format csv -- python3 $R/tools/prof_driver.py $W $REPS > $OUT/${W}_fetch.log 2>&1
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/${W}_write -o write --output-format csv -- python3 $R/tools/prof_driver.py $W $REPS > $OUT/${W}_write.log 2>&1
    timeout -s KILL 120 rocprofv3 --pmc $VALU -d $OUT/${W}_valu -o valu --output-format csv -- python3 $R/tools/prof_driver.py $W $REPS > $OUT/${W}_valu.log 2>&1
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_kt -o kt -- python3 $R/tools/prof_driver.py $W $REPS > $OUT/${W}_kt.log 2>&1
    if [ $W = ntt22 ] || [ $W = ntt24 ]; then
        timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/${W}_grbm -o grbm --output-format csv -- python3 $R/tools/prof_driver.py $W $REPS > $OUT/${W}_grbm.log 2>&1
    fi
done
if [ -x $R/tools/_build/hash_bench ]; then timeout -k 10 60 $R/tools/_build/hash_bench > $OUT/hash_bench.txt 2>&1; fi
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
ls $OUT

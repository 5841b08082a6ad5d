set -e
R=$PWD
OUT=$R/gpurun_out/mk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/valu -o valu --output-format csv -- python3 $R/tools/prof_driver.py merkle24 3 > $OUT/valu.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F32 -d $OUT/misc -o misc --output-format csv -- python3 $R/tools/prof_driver.py merkle24 3 > $OUT/misc.log 2>&1 || true
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/tools/prof_driver.py merkle24 3 > $OUT/kt.log 2>&1
ls $OUT/*

#!/bin/bash
# MFMA-pipe utilisation of the dominant conv kernel: SQ counters of the standalone harness (one kernel, 4096 boards,
# random data).  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles the matrix pipe is busy (summed over SIMDs);
# GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Run on the GPU box from the repo root.
set -e
OUT=$PWD/gpurun_out/prof_${1:-mfma}
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o mfma -- $REPO/tools/conv_bench 4096 10 5 1 0 > $OUT/run.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- $REPO/tools/conv_bench 4096 10 5 1 0 >> $OUT/run.log 2>&1
ls $OUT/pmc $OUT/trace

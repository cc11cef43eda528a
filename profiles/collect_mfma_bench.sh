#!/bin/bash
# Matrix-pipe counters of the conv kernel inside the self-play bench itself (2 rounds; real activations, board groups).
TAG=${1:-r2m}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o m -- python3 $REPO/bench.py --steps 1 --warmup 0 --rounds-per-step 2 --no-cpu-baseline --no-other-format --no-tower-parity --no-exchange > $OUT/bench.json 2> $OUT/pmc.err || { echo "pmc pass failed"; tail -c 300 $OUT/pmc.err; exit 1; }
cd $REPO && python profiles/pmc_summary.py $OUT/pmc/m_counter_collection.csv | grep -E "kernel|duo"

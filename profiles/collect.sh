#!/bin/bash
# Run on the GPU box from the repo root: kernel trace + two PMC passes (HBM read / write bytes) of the
# default bench workload, short (10 steps).  Outputs under gpurun_out/prof_$1.
set -e
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/write_bench.json 2> $OUT/write.err
ls -R $OUT | head -40

#!/bin/bash
# HBM-side counters of the grouped tower on a short run (2 rounds): the counter passes of profiles/collect.sh abort inside the
# profiler on the full-length run.  Outputs under gpurun_out/prof_$1.
TAG=${1:-r2s}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 1 --warmup 0 --rounds-per-step 2 --no-cpu-baseline --no-tower-parity --no-exchange --no-other-format"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err || { echo "fetch pass failed"; tail -c 400 $OUT/fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/write_bench.json 2> $OUT/write.err || { echo "write pass failed"; exit 1; }
ls -la $OUT/pmc_fetch $OUT/pmc_write

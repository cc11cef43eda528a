#!/bin/bash
# Run on the GPU box from the repo root: kernel trace of the default bench workload, short (2 steps of 41 rounds), and, with
# PMC=1, two counter passes (HBM read / write bytes) of the same run - with the tower in groups of boards the profiler aborts in
# those (eleven times the dispatches); profiles/collect_pmc_short.sh takes them on a 2-round run.  Outputs under gpurun_out/prof_$1.
set -e
TAG=${1:-r2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tower-parity --no-exchange --no-other-format"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
[ "$PMC" = 1 ] || { ls -R $OUT | head -20; exit 0; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/write_bench.json 2> $OUT/write.err
ls -R $OUT | head -40

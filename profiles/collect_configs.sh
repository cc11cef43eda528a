#!/bin/bash
# Kernel traces of the smaller configurations (run on the GPU box from the repo root).
set -e
OUT=$PWD/gpurun_out/prof_cfg
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/$name -o t -- python3 $REPO/bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-other-format > $OUT/$name.json 2> $OUT/$name.err; }
run reversi --game reversi --board 8 --sims 400 --blocks 6
run go9 --game go --board 9 --sims 400 --blocks 10
run gomoku --game mnk --k 5 --board 15 --sims 800 --blocks 10
ls $OUT

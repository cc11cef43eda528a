#!/bin/bash
# Round-3 measurements, run on the GPU box from the repo root; every step under its own time limit, the script stops at the first
# failure.  Outputs under gpurun_out/r3p (kernel trace and counter passes under gpurun_out/prof_r3, prof_r3s as profiles/refresh.py expects).
set -e
OUT=$PWD/gpurun_out/r3p
mkdir -p $OUT
# 1. kernel trace of the default workload (short) and the two HBM counter passes (2 rounds: counter passes are kept to a few thousand dispatches)
bash profiles/collect.sh r3
bash profiles/collect_pmc_short.sh r3s
# 2. the same bench with the half-precision tower
timeout -k 10 300 python3 bench.py --dtype f16 --no-cpu-baseline --no-other-format > $OUT/bench_f16.json 2> $OUT/bench_f16.err
# 3. one rank under a launcher's environment: the RCCL process group, broadcast, all-reduce and barrier on a real device
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-format --no-tower-parity > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
# 4. the plain two-rank command, both ranks on this one card (RCCL refuses two ranks per device: gloo carries the exchanges)
AZ_BENCH_DEVICE=0 AZ_BENCH_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --games 1452 --steps 4 --warmup 1 --no-tower-parity > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err
# 5. the smaller configurations, default length (every game completes a search and a move)
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-other-format --no-tower-parity --no-exchange > $OUT/cfg_$name.json 2> $OUT/cfg_$name.err; }
run ttt --game mnk --k 3 --board 3 --sims 100 --blocks 2
run reversi --game reversi --board 8 --sims 400 --blocks 6
run go9 --game go --board 9 --sims 400 --blocks 10
run gomoku --game mnk --k 5 --board 15 --sims 800 --blocks 10
ls $OUT

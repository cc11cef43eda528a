#!/bin/bash
# A/B of the two-workgroups-per-CU conv kernel on boards whose slab needs 27 pieces (AZ_CONV_DUO=0: one-workgroup slab kernel).
out=gpurun_out/duo_ab_boards.log
: > $out
run() {
  for d in 0 1 0 1; do
    echo "== $* AZ_CONV_DUO=$d" >> $out
    AZ_CONV_DUO=$d timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-format --no-tower-parity --no-exchange "$@" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print('moves/s %.1f  ms/round %.3f  conv %.1f TFLOP/s  avg launch %.4f ms  boards/launch %.0f' % (d['value'], d['ms_per_round'], r['achieved'], r['avg_launch_ms'], r.get('boards_per_launch', 0)))
" >> $out || exit 1
  done
}
run --game go --board 9 --sims 400 --blocks 10 --games 4045
run --game mnk --board 15 --k 5 --sims 800 --blocks 10 --games 4074
run --game reversi --board 8 --sims 400 --blocks 6 --games 4096
cat $out

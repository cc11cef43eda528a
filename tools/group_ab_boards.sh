#!/bin/bash
# A/B of the tower's board groups on the other named configurations (whole batch per layer: -1, automatic groups: 0).
out=gpurun_out/group_ab_boards.log
: > $out
run() {
  for g in -1 0 -1 0; do
    echo "== $* AZ_TOWER_GROUP_BOARDS=$g" >> $out
    AZ_TOWER_GROUP_BOARDS=$g timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-format --no-tower-parity --no-exchange "$@" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print('moves/s %.1f  ms/round %.3f  conv %.1f TFLOP/s  avg launch %.4f ms  boards/launch %.0f' % (d['value'], d['ms_per_round'], r['achieved'], r['avg_launch_ms'], r.get('boards_per_launch', 0)))
" >> $out || exit 1
  done
}
run --game go --board 9 --sims 400 --blocks 10 --games ${GO9:-4045}
run --game mnk --board 15 --k 5 --sims 800 --blocks 10 --games ${GOMOKU:-4074}
run --game reversi --board 8 --sims 400 --blocks 6 --games ${REVERSI:-4096}
cat $out

#!/bin/bash
# A/B of the tower's board-group size in the self-play bench, same box, back to back: one argument per run, the boards per
# group (AZ_TOWER_GROUP_BOARDS: -1 = whole batch per layer, 0 = automatic, N = groups of N boards).
out=gpurun_out/group_sweep.log
: > $out
for g in "$@"; do
  echo "== AZ_TOWER_GROUP_BOARDS=$g" >> $out
  AZ_TOWER_GROUP_BOARDS=$g timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-format 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print('moves/s %.2f  ms/round %.2f  conv %.1f TFLOP/s  avg launch %.4f ms  launches %d  flops/launch %.4g' % (d['value'], d['ms_per_round'], r['achieved'], r['avg_launch_ms'], r['launches'], r['flops_per_launch']))
" >> $out || exit 1
done
cat $out

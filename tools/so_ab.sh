#!/bin/bash
# A/B of two builds of the library in the self-play bench on one box: tools/prev/libazamd.so (old) against tools/prev/libazamd_new.so.
out=gpurun_out/so_ab.log
: > $out
for v in old new old new; do
  if [ $v = old ]; then cp tools/prev/libazamd.so alphazero_amd/csrc/libazamd.so; else cp tools/prev/libazamd_new.so alphazero_amd/csrc/libazamd.so; fi
  echo "== $v" >> $out
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-format "$@" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print('moves/s %.2f  ms/round %.3f  conv %.1f TFLOP/s  phases %s' % (d['value'], d['ms_per_round'], r['achieved'], d['phase_ms_per_round']))
" >> $out || exit 1
done
cp tools/prev/libazamd_new.so alphazero_amd/csrc/libazamd.so
cat $out

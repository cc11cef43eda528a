#!/bin/bash
# bf16 against half precision in the self-play bench, same box, alternating runs (the conv kernels are the same template;
# what differs is the operand format the matrix cores see).  Output: gpurun_out/f16_ab.log
out=gpurun_out/f16_ab.log
: > $out
for dt in bf16 f16 bf16 f16; do
  echo "== --dtype $dt" >> $out
  timeout -k 10 200 python3 bench.py --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --no-other-format --no-tower-parity --no-exchange 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']; t=(d.get('gpu_telemetry') or {}).get('samples') or [{}]
        print('sims/s %.0f  ms/round %.2f  conv %.1f TFLOP/s  avg launch %.4f ms  socket power %s W  gfxclk %s' % (d['sims_per_sec'], d['ms_per_round'], r['achieved'], r['avg_launch_ms'], t[-1].get('socket_power_w'), (t[-1].get('gfxclk_mhz') or [None])[0]))
" >> $out || exit 1
done
cat $out

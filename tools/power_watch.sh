#!/bin/bash
# Samples the GPU's power draw, clocks and throttle-residency counters (rocm-smi, read-only queries) while the self-play bench runs.
#   tools/power_watch.sh [bench arguments]
out=gpurun_out/power_watch.log
: > $out
rocm-smi --showmaxpower 2>&1 | grep -i "power" >> $out
echo "== metrics before" >> $out
timeout 20 rocm-smi --showmetrics 2>&1 | grep -iE "power|residency|throttle|gfxclk|current_gfx|average_gfx|temperature_hotspot|energy|activity" >> $out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-format --no-tower-parity --no-exchange "$@" > gpurun_out/power_bench.log 2>&1 &
pid=$!
sleep 12
for i in 1 2 3; do
  kill -0 $pid 2>/dev/null || break
  echo "== metrics during the run, sample $i" >> $out
  timeout 20 rocm-smi --showmetrics 2>&1 | grep -iE "power|residency|throttle|gfxclk|current_gfx|average_gfx|temperature_hotspot|energy|activity" >> $out
  timeout 10 rocm-smi -P 2>&1 | grep -v "^=\|^$" >> $out
  sleep 4
done
wait $pid
echo "== metrics after" >> $out
timeout 20 rocm-smi --showmetrics 2>&1 | grep -iE "power|residency|throttle|energy" >> $out
cat $out

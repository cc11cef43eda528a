#!/bin/bash
# cycles and stall counters of the two-workgroup conv kernel's ablation builds (tools/cb_a0 base, cb_a2 without the weights' LDS-DMA)
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  RELU=1 timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/abl6_$v -o p -- $R/tools/cb_$v 3993 10 6 1 0 > $R/gpurun_out/abl6_$v.log 2>&1 || exit 1
  RELU=1 timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl6t_$v -o t -- $R/tools/cb_$v 3993 10 6 1 0 > $R/gpurun_out/abl6t_$v.log 2>&1 || exit 1
done
cd $R
for v in "$@"; do echo "== $v"; python profiles/pmc_summary.py gpurun_out/abl6_$v/p_counter_collection.csv | grep duo; python - <<PY
import csv
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open("gpurun_out/abl6t_$v/t_kernel_trace.csv")) if "duo" in r["Kernel_Name"]]
print("kernel trace: %d launches, mean %.1f us" % (len(d), sum(d)/len(d)/1e3))
PY
done

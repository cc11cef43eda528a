#!/bin/bash
# Ablation table of the slab conv kernel on the GPU box (run from the repo root): harness builds with parts of the K loop
# removed (tools/conv_bench.hip compiled with -DSTAMP and the ABL_* flags, see tools/conv_ablation_kernels.h), 4096 boards of 19x19,
# residual variant, on uniform random operands and on ReLU-like operands (RELU=1: half zeros, rest positive).  Each line:
# wall TFLOP/s, and the in-kernel clock / cycles per 32-MFMA interval of one workgroup from shader-clock stamps.
OUT=gpurun_out/ablation_${1:-r2}.txt
: > $OUT
for v in base nt noread nodma noread_nodma mfmaonly; do
  for data in random relu; do
    if [ $data = relu ]; then export RELU=1; else unset RELU; fi
    echo "== $v ($data operands)" >> $OUT
    timeout -k 10 120 tools/cb_$v 4096 30 5 1 0 2>&1 | grep -E "K loop|variant" >> $OUT || exit 1
  done
done
cat $OUT

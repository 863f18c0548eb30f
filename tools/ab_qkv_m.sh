#!/bin/bash
# QKV GEMM at smaller row counts (strong scaling: 350720 / N rows per rank): weights-in-registers kernel (threshold lowered) vs the
# kernels the dispatcher takes below its 65,536-row threshold
for M in 175360 87680 43840 21920 10960 5480; do
  echo "== M=$M wreg"; MST_GEMM_WREG_MIN_M=1 timeout -k 5 120 python tools/bench_gemm.py $M qkv 2>/dev/null | tail -1
  echo "== M=$M default"; timeout -k 5 120 python tools/bench_gemm.py $M qkv 2>/dev/null | tail -1
done

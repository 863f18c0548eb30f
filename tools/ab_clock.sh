#!/bin/bash
# Steady-state time and in-kernel clock of the QKV GEMM: the mid-tile kernel (no stamps), then every -DWREG_CLOCK variant.
echo "== mid (MST_GEMM_WREG=0)"; MST_GEMM_WREG=0 timeout -k 5 120 python tools/wreg_clock.py 2>/dev/null | tail -1
for round in 1 2; do
for lib in new-vit_amd/mst/hip/libq_CLK*.so; do
  echo "== round $round $lib"
  MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/wreg_clock.py 2>/dev/null | tail -1
done
done

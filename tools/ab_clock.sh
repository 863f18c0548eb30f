#!/bin/bash
# Steady-state time (2 s of launches first) of the QKV GEMM under the default library and every libq_*.so variant; -DWREG_CLOCK
# builds also report the in-kernel clock.  Correctness of each variant first (tools/check_qkv.py).
for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/libq_*.so; do
  [ -e "$lib" ] || continue
  echo "== check $lib"; MST_HIP_LIB=$PWD/$lib timeout -k 5 300 python tools/check_qkv.py 2>&1 | grep -c "'ok': True"
done
for round in 1 2; do
for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/libq_*.so; do
  [ -e "$lib" ] || continue
  echo "== round $round $lib"
  MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/wreg_clock.py 2>/dev/null | tail -1
done
done

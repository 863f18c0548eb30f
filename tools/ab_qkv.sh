#!/bin/bash
# QKV-shaped mst_gemm: correctness (tools/check_qkv.py) then timing (tools/bench_gemm.py) of the default build, of the build with
# the weights-in-registers kernel switched off (MST_GEMM_WREG=0 -> k_gemm16_mid.hip) and of every variant library libq_*.so.
mkdir -p gpurun_out
echo "== check default"; timeout -k 5 300 python tools/check_qkv.py 2>&1 | tail -7 || exit 1
for round in 1 2; do
  echo "== round $round wreg"; timeout -k 5 120 python tools/bench_gemm.py 350720 qkv 2>/dev/null | tail -1
  echo "== round $round mid";  MST_GEMM_WREG=0 timeout -k 5 120 python tools/bench_gemm.py 350720 qkv 2>/dev/null | tail -1
  for lib in new-vit_amd/mst/hip/libq_*.so; do
    [ -e "$lib" ] || continue
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/bench_gemm.py 350720 qkv 2>/dev/null | tail -1
  done
done

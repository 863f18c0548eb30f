#!/bin/bash
# Per library variant of the single-role block kernel: parity (tests) then tools/bench_block.py (same box, back to back, two rounds).
mkdir -p gpurun_out
for lib in new-vit_amd/mst/hip/libv_*.so; do
  case $lib in *STAMPS*|*NO*) continue;; esac
  echo "== check $lib"; MST_HIP_LIB=$PWD/$lib timeout -k 5 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k block_fused_single_role 2>&1 | tail -1
done
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/libv_*.so; do
    case $lib in *STAMPS*) continue;; esac
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/bench_block.py 2>/dev/null | grep "^single_blk"
  done
done

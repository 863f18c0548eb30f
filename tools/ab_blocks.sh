#!/bin/bash
# Run tools/bench_block.py once per library variant of the single-role block kernel (same box, back to back, two rounds).
mkdir -p gpurun_out
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/libv_*.so; do
    case $lib in *STAMPS*) continue;; esac
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/bench_block.py 2>/dev/null | grep "^single_blk"
  done
done

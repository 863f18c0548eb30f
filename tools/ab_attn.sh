#!/bin/bash
# Attention at the bench shape under the default library and every variant new-vit_amd/mst/hip/liba_*.so (two rounds).
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/liba_*.so; do
    [ -e "$lib" ] || continue
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/bench_attn.py 2>/dev/null | tail -1
  done
done

#!/bin/bash
# Attention at the bench shape under the default library and every variant new-vit_amd/mst/hip/liba_*.so: correctness of the
# non-ablation variants first (tools/check_attn.py), then two rounds of tools/bench_attn.py.
for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/liba_*.so; do
  [ -e "$lib" ] || continue
  case $lib in *NO_*) continue;; esac
  echo "== check $lib"; MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/check_attn.py 2>&1 | tail -4
done
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/liba_*.so; do
    [ -e "$lib" ] || continue
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/bench_attn.py 2>/dev/null | tail -1
  done
done

#!/bin/bash
# 4-wave vs 8-wave workgroups at a token count both tile without idle waves (N = 1280 = 10 x 128 = 5 x 256)
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/liba_WG8.so; do
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 120 python tools/bench_attn.py 1280 2>/dev/null | tail -1
  done
done

#!/bin/bash
# Whole bench step (bench.py, 10 steps) under the default library and every variant new-vit_amd/mst/hip/libv_*.so, two rounds.
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/libv_*.so; do
    [ -e "$lib" ] || continue
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['parity_check']['max_abs_err'])"
  done
done

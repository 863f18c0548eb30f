#!/bin/bash
# bench.py (no CPU baseline) once per library variant new-vit_amd/mst/hip/libv_*.so and the default, two rounds back to back on one box.
for round in 1 2; do
  for lib in new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/libv_*.so; do
    [ -e "$lib" ] || continue
    echo "== round $round $lib"
    MST_HIP_LIB=$PWD/$lib timeout -k 5 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  done
done

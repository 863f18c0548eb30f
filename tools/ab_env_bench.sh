#!/bin/bash
# Whole bench step with an environment switch off / on (two rounds).  Usage: tools/ab_env_bench.sh VAR   (VAR=0 vs unset)
VAR=${1:?variable}
for round in 1 2; do
  for val in 0 ""; do
    echo "== round $round $VAR=${val:-<unset>}"
    if [ -n "$val" ]; then export $VAR=$val; else unset $VAR; fi
    timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['parity_check']['max_abs_err'])"
  done
done

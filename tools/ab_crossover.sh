#!/bin/bash
# forward latency (ms) vs tokens per call at 224^2: fused pipeline forced on / off (MST_FUSED_MIN_TOKENS is read once per process)
for v in 0 1000000000000; do
  echo "== MST_FUSED_MIN_TOKENS=$v"; MST_FUSED_MIN_TOKENS=$v timeout -k 5 200 python tools/bench_crossover.py 2>/dev/null | tail -1
done

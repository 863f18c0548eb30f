#!/bin/bash
# HBM traffic counters for the bench kernels: separate rocprofv3 --pmc passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2).
# Usage (GPU box): bash tools/profile_pmc.sh <tag>
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o $C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/${C}_stdout.json 2> $OUT/${C}_stderr.log
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
# prune bulky raw outputs (gpurun copies back at most 64 MiB): keep the summaries, the per-kernel stats and the bench line
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" -o -name "*.pftrace" \) -delete 2>/dev/null || true

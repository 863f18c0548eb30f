#!/bin/bash
# SQ / GRBM counters of the bench kernels (one rocprofv3 --pmc pass: 8 SQ slots + GRBM), for MFMA-pipe utilisation and
# the split of wave time into issue / issue-stall / parked.  Usage (GPU box): bash tools/profile_sq.sh <tag>
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/stdout.json 2> $OUT/stderr.log
python3 $GRAFT_REPO_ROOT/tools/sq_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
# prune bulky raw outputs (gpurun copies back at most 64 MiB): keep the summaries, the per-kernel stats and the bench line
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" -o -name "*.pftrace" \) -delete 2>/dev/null || true

#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (run on the GPU box via gpurun).
# Usage: bash tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv to profiles/
set -e -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
ls -R $OUT | head -30
# prune bulky raw outputs (gpurun copies back at most 64 MiB): keep the summaries, the per-kernel stats and the bench line
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" -o -name "*.pftrace" \) -delete 2>/dev/null || true

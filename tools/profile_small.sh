#!/bin/bash
# rocprofv3 kernel stats of the small-shape forward (tools/bench_small.py: 1 x 16 x 224^2 and 1 x 32 x 224^2, bf16 and fp32).  GPU box: bash tools/profile_small.sh <tag>
TAG=${1:-x}
OUT=$PWD/gpurun_out/small_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sm -- python3 $GRAFT_REPO_ROOT/tools/bench_small.py --bf16-16-only > $OUT/stdout.log 2> $OUT/stderr.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/sm_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in rows[:22]:
    print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.1f} {r["Percentage"]}%')
PY
tail -2 $OUT/stdout.log
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*agent_info.csv" \) -delete 2>/dev/null || true

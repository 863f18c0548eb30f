#!/bin/bash
# rocprofv3 kernel stats of the DINOv2 training step (tools/bench_train.py, first case only).  Usage: bash tools/profile_train.sh <tag>
TAG=${1:-x}
OUT=$PWD/gpurun_out/train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py ${2:---only-dino-c1} $3 $4 $5 > $OUT/stdout.log 2> $OUT/stderr.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/train_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in rows[:14]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.1f} {r["Percentage"]}%')
PY
tail -2 $OUT/stdout.log
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*agent_info.csv" \) -delete 2>/dev/null || true

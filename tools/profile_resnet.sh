#!/bin/bash
# rocprofv3 kernel stats of the ResNetSliceTrans(34) forward at the configs[3] shape.  Usage (GPU box): bash tools/profile_resnet.sh <tag>
TAG=${1:-x}
OUT=$PWD/gpurun_out/resnet_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o rn -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only-model > $OUT/stdout.log 2> $OUT/stderr.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/rn_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in rows[:16]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.1f} {r["Percentage"]}%')
PY
tail -2 $OUT/stdout.log
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*agent_info.csv" \) -delete 2>/dev/null || true

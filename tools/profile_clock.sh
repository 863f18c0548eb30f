#!/bin/bash
# Effective shader clock per kernel of the bench step: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration.
# Usage (GPU box): bash tools/profile_clock.sh <tag>
TAG=${1:-x}
OUT=$PWD/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o clk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/stdout.json 2> $OUT/stderr.log
python3 - <<PY
import csv, glob, collections
out = "$OUT"
cc = glob.glob(out + "/*counter_collection.csv")[0]
kt = glob.glob(out + "/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    name, ns = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
    if ns <= 0: continue
    a = agg[name.split("(")[0][:60]]
    a[0] += float(r["Counter_Value"]) / 8.0
    a[1] += ns
    a[2] += 1
for k, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / n > 50000:
        print(f"{k:60s} launches {n:4d} avg_us {ns / n / 1e3:8.1f} clock_GHz {cyc / ns:5.3f}")
PY

#!/bin/bash
# LDS bank-conflict counters of the bench kernels (one rocprofv3 --pmc pass).  Usage (GPU box): bash tools/profile_lds.sh <tag>
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o lds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/stdout.json 2> $OUT/stderr.log
python3 - <<PY
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        a = acc[(r.get("Kernel_Name") or "")[:70]][r.get("Counter_Name")]
        a[0] += float(r.get("Counter_Value") or 0); a[1] += 1
res = {}
for k, cs in acc.items():
    avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    if avg.get("SQ_INSTS_LDS", 0) < 1e5: continue
    act = max(avg.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    res[k] = {"lds_insts": round(avg.get("SQ_INSTS_LDS", 0)), "bank_conflict_cycles_per_active_cycle": round(avg.get("SQ_LDS_BANK_CONFLICT", 0) / act, 4),
              "addr_conflict": round(avg.get("SQ_LDS_ADDR_CONFLICT", 0)), "unaligned_stall": round(avg.get("SQ_LDS_UNALIGNED_STALL", 0))}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

#!/bin/bash
# LDS / instruction-mix counters of the bench kernels (second SQ pass): bank conflicts, LDS-array cycles, LDS issue stalls, instruction counts.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/stdout.json 2> $OUT/stderr.log
python3 - $OUT > $OUT/summary.json <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        a = acc[(r.get("Kernel_Name") or "")[:70]][r.get("Counter_Name")]
        a[0] += float(r.get("Counter_Value") or 0); a[1] += 1
out = {}
for k, cs in acc.items():
    v = {c: x[0] / max(x[1], 1) for c, x in cs.items()}
    if v.get("GRBM_GUI_ACTIVE", 0) / 8 < 1e5:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    out[k] = {"launches": int(max(x[1] for x in cs.values())), "kernel_cycles": round(cyc),
              "lds_array_busy_frac_per_cu": round(v.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, 3),
              "lds_bank_conflict_cycles_per_cu": round(v.get("SQ_LDS_BANK_CONFLICT", 0) / 256),
              "lds_insts_per_simd": round(v.get("SQ_INSTS_LDS", 0) / 1024), "valu_insts_per_simd": round(v.get("SQ_INSTS_VALU", 0) / 1024),
              "mfma_insts_per_simd": round(v.get("SQ_INSTS_MFMA", 0) / 1024)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["launches"])[:8]), indent=1))
PY
cat $OUT/summary.json
# prune bulky raw outputs (gpurun copies back at most 64 MiB): keep the summaries, the per-kernel stats and the bench line
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" -o -name "*.pftrace" \) -delete 2>/dev/null || true

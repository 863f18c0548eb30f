#!/bin/bash
# SQ counters of the block kernels (old producer/consumer, single-role) on tools/bench_block.py; two --pmc passes.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/sqb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/p1 -o sq -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py > $OUT/p1_stdout.log 2> $OUT/p1_stderr.log
python3 $GRAFT_REPO_ROOT/tools/sq_summary.py $OUT/p1 > $OUT/p1_summary.json
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/p2 -o sq -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py > $OUT/p2_stdout.log 2> $OUT/p2_stderr.log
python3 - $OUT/p2 > $OUT/p2_summary.json <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        a = acc[(r.get("Kernel_Name") or "")[:80]][r.get("Counter_Name")]
        a[0] += float(r.get("Counter_Value") or 0); a[1] += 1
print(json.dumps({k: {c: round(v[0] / max(v[1], 1)) for c, v in cs.items()} for k, cs in acc.items() if "block16" in k or "mlp16" in k}, indent=1))
PY
cat $OUT/p1_summary.json $OUT/p2_summary.json
rm -rf $OUT/p1/*/*.db $OUT/p2/*/*.db 2>/dev/null

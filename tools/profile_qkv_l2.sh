#!/bin/bash
# L2 behaviour of the QKV GEMM (tools/bench_qkv.py): hits / misses / requests, one rocprofv3 --pmc pass per counter.
# Usage (GPU box): bash tools/profile_qkv_l2.sh <tag>   (library chosen by MST_HIP_LIB)
TAG=${1:-x}
OUT=$PWD/gpurun_out/qkvl2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o $C -- python3 $GRAFT_REPO_ROOT/tools/bench_qkv.py > $OUT/${C}.log 2>&1 || echo "$C failed"
done
python3 - <<PY
import csv, glob, os
out = "$OUT"
for f in sorted(glob.glob(out + "/*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
    if not rows: continue
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        print("$TAG", k, "per launch", round(sum(v) / len(v)), "launches", len(v))
PY

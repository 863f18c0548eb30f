#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the QKV GEMM for whatever library MST_HIP_LIB / MST_GEMM_BIG select.  Usage: bash tools/profile_qkv.sh <tag>
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/qkv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o $C -- python3 $GRAFT_REPO_ROOT/tools/bench_qkv.py > $OUT/${C}.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'gemm' in k: print('$TAG', k[:40], 'fetch MB', round(v['fetch_bytes_per_launch']/1e6), 'write MB', round(v['write_bytes_per_launch']/1e6))"

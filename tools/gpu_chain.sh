#!/bin/bash
# Run GPU steps one after another; a step that times out or is killed ends the chain (no further GPU step in the call).
# usage: tools/gpu_chain.sh "<secs> <logname> <command...>" ...
mkdir -p gpurun_out
status=0
for spec in "$@"; do
    secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
    echo "=== [$name] $cmd" | tee -a gpurun_out/chain.log
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] rc=$rc" | tee -a gpurun_out/chain.log
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / killed: stopping the chain"; exit $rc; fi
    [ $rc -ne 0 ] && status=$rc
done
exit $status

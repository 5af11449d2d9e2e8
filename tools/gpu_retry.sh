#!/bin/bash
# Dev helper (build container): gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).
# usage: tools/gpu_retry.sh TIMEOUT 'command'
T=$1; shift
for attempt in $(seq 1 20); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 60
done
exit 3

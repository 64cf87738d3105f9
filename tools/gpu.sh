#!/bin/bash
# tools/gpu.sh <tag> <timeout> '<command>': one gpurun call; when no GPU slot / box is free (exit 3: nothing ran, nothing was
# charged) it asks again after two minutes, up to 10 times.  A command that RAN is never repeated.  Log: gpurun_out/<tag>/call.log
tag=$1; to=$2; shift 2
mkdir -p gpurun_out/$tag
for i in 1 2 3 4 5 6 7 8 9 10; do
    /usr/local/graft/bin/gpurun --timeout $to -- "$@" > gpurun_out/$tag/call.log 2>&1
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 120
done
exit 3

#!/bin/bash
# tools/gpu.sh, retried while the pod has no free GPU slot (exit code 3: nothing ran, nothing was charged)
# usage: tools/gpu_retry.sh <timeout-seconds> '<command>' <logfile>
for i in 1 2 3 4 5 6 7 8; do
  tools/gpu.sh "$1" "$2" > "$3" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3

#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout> <script> <logfile>   — retries while the pod's GPU slots are busy (exit 3)
for i in $(seq 1 20); do
  gpurun --timeout "$1" -- "bash $2" > "$3" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3

#!/bin/bash
# gpurun with retries while the pool is busy (exit 3 / "transient"): usage tools/gpurun_retry.sh <timeout> '<command>'
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out" | tail -${3:-80}; exit $rc
done
echo "gpurun: still busy after 20 attempts"; exit 3

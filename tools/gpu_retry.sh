#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy: tools/gpu_retry.sh <timeout> <logfile> <command...>
T=$1; LOG=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  grep -q "status=transient" $LOG || break
  sleep 45
done

#!/bin/bash
# gp.sh <log> <timeout> <command...>: one gpurun call, tried again (every 2 minutes, at most 12 times) only while the pool says "no
# slot / no box right now" (exit code 3: nothing ran, nothing was charged)
log=$1; to=$2; shift 2
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3

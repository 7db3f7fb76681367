#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT 'command'   — gpurun, retried while the pod's GPU slots are busy (exit 3 / "transient": nothing charged)
T=$1; shift
for i in $(seq 1 30); do
  OUT=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); RC=$?
  if echo "$OUT" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$OUT"; exit $RC
done
echo "$OUT"; exit 3

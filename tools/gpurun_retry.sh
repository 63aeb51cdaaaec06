#!/bin/bash
# Calls gpurun, retrying only while the pool reports "no box free" (exit 3 / transient); the
# command itself is never retried after it has run.
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out"; exit $rc
done
echo "gpurun: still no box after 30 tries"; exit 3

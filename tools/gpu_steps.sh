#!/bin/bash
# Run GPU steps one after another; a step killed at its limit (rc >= 124) ends the whole call (no further GPU step).
# usage: source tools/gpu_steps.sh; step <seconds> <logfile> <cmd...>
step() {
  local limit=$1 log=$2; shift 2
  echo "== step: $* (limit ${limit}s)"
  timeout -k 10 "$limit" "$@" > "$log" 2> "${log%.*}.err"
  local rc=$?
  echo "== rc=$rc  ($log)"
  if [ $rc -ge 124 ]; then echo "STEP KILLED rc=$rc: stopping"; exit $rc; fi
  return 0
}

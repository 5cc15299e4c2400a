#!/bin/bash
# N consecutive runs of a pytest selection, one process each, stopping at the first failure (no GPU step after a fault).
# usage: tools/soak.sh <log> <n> <pytest args...>
log=$1; n=$2; shift 2
mkdir -p "$(dirname "$log")"; : > "$log"
for i in $(seq 1 "$n"); do
  echo "=== soak run $i $(date +%H:%M:%S): pytest $*" | tee -a "$log"
  timeout -k 10 900 python -m pytest "$@" -x -q >> "$log" 2>&1
  rc=$?
  tail -n 1 "$log"
  if [ $rc -ne 0 ]; then echo "=== soak run $i FAILED rc=$rc" | tee -a "$log"; exit $rc; fi
done
echo "=== $n of $n runs passed" | tee -a "$log"

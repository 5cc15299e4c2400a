#!/bin/bash
# One variant per PROCESS (identical allocation sequence in each), alternating, three times: the decisive form of an A/B on this chip, where
# identical handles of one process can differ by 5 % (tools/placement_probe.py).   tools/process_ab.sh <size> <interp> <angle step> VAR=VAL ... ("-" = default)
size=$1; interp=$2; step=$3; shift 3
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then python3 tools/march_ab.py --size $size --interp $interp --flags 0 --angles 0 180 $step --rounds 2 2>&1 | grep -v amdgpu | sed "s/env=-   /env=default/"
    else env $v python3 tools/march_ab.py --size $size --interp $interp --flags 0 --angles 0 180 $step --rounds 2 2>&1 | grep -v amdgpu | sed "s#env=-  *#env=$v #"; fi
  done
done

#!/bin/bash
# A/B on one box: tools/ab.sh <libA.so> <libB.so> <prof_case args...>   (alternates 3x; prints ms/launch)
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    echo -n "$(basename $(dirname $L)) : "
    VT_LIB=$L python3 tools/prof_case.py "$@" 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-60
  done
done

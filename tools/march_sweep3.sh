#!/bin/bash
size=${1:-1024}
for cfg in "0 2" "0 4" "0 8" "4 4" "4 8" "0 16"; do set -- $cfg
  echo -n "VT_TILE=$1 VT_DCH=$2 : "
  VT_TILE=$1 VT_DCH=$2 python3 tools/prof_case.py --size $size --interp linear --angle 45 --iters 5 2>&1 | grep -v amdgpu.ids | cut -c50-180
done

#!/bin/bash
# lookahead sweep: tools/march_sweep4.sh <size> <interp> <angle>
size=${1:-1024}; interp=${2:-linear}; angle=${3:-45}
for t in 0 4; do for la in 1 2 3 4 6; do for d in 16 32; do
  echo -n "VT_TILE=$t VT_LA=$la VT_DCH=$d : "
  VT_TILE=$t VT_LA=$la VT_DCH=$d python3 tools/prof_case.py --size $size --interp $interp --angle $angle --iters 10 2>&1 | grep -v amdgpu.ids | cut -c50-180
done; done; done

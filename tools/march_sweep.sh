#!/bin/bash
# marching-kernel configuration sweep: tools/march_sweep.sh <size> <interp> <angle>
size=${1:-1024}; interp=${2:-linear}; angle=${3:-45}
for t in 0 2 3 4; do
  for d in 16 32 64; do
    echo -n "VT_TILE=$t VT_DCH=$d : "
    VT_TILE=$t VT_DCH=$d python3 tools/prof_case.py --size $size --interp $interp --angle $angle --iters 5 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done
done

#!/bin/bash
# bounding-box kernel, general rotation: which tile shape?  tools/tile_sweep.sh <interp> <size>
interp=${1:-bspline}; size=${2:-384}
for t in 0 1 2 3 4; do
  echo -n "VT_TILE=$t : "
  VT_TILE=$t python3 tools/prof_case.py --size $size --interp $interp --general --flags 128 --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-150
done

#!/bin/bash
for interp in linear bspline; do for s in 512 384; do for o in 0 1 0 1; do
  echo -n "$interp $s order=$o : "
  VT_TILE_ORDER=$o python3 tools/prof_case.py --size $s --interp $interp --general --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-70
done; done; done

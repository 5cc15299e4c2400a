#!/bin/bash
# quick timing sweep over marching configs / angles (run on the GPU box through gpurun)
for interp in linear filt_bspline; do
  for tile in 0 1 2 3; do
    for ang in 0 20 45; do
      VT_TILE=$tile python3 tools/prof_case.py --size 512 --interp $interp --angle $ang --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg=$tile /"
    done
  done
done

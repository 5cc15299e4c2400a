#!/bin/bash
for s in 512 1024; do for a in 0 45; do for t in 0 4 5 0 4 5; do
  echo -n "size=$s angle=$a VT_TILE=$t : "
  VT_TILE=$t python3 tools/prof_case.py --size $s --interp filt_bspline --angle $a --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-130
done; done; done

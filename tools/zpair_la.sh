#!/bin/bash
for s in 512 1024; do for a in 0 45; do for la in 1 2 3 1 2 3; do
  echo -n "size=$s angle=$a VT_LA=$la : "
  VT_LA=$la python3 tools/prof_case.py --size $s --interp filt_bspline --angle $a --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-150
done; done; done

#!/bin/bash
size=${1:-512}
for a in 0 20 45; do for d in 32 64; do
  echo -n "angle=$a VT_DCH=$d : "
  VT_DCH=$d python3 tools/prof_case.py --size $size --interp filt_bspline --angle $a --iters 10 2>&1 | grep -v amdgpu.ids | cut -c55-200
done; done

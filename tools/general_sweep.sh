#!/bin/bash
# general (non-separable) matrices: packed-footprint kernel (flags 16 = NO_ZSEP) vs bounding-box kernel (16+128)
for interp in linear bspline; do
  for c in rot_general rot_scale_shift shear mirror minify shift_frac; do
    for f in 16 144 272; do python3 tools/prof_case.py --size 512 --interp $interp --case $c --iters 10 --flags $f 2>&1 | grep -v amdgpu.ids; done
  done
done

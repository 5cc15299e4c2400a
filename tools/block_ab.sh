#!/bin/bash
# general-rotation kernels, 512^3 (25,-40,70) sxyz: lane-block kernel (cubic by default; trilinear through VT_BLOCK_LINEAR) against the
# round-1 box / packed kernels, one process each (for an in-process comparison over 100 random rotations: tools/general_ab.py)
set -e
for interp in linear bspline; do
  echo "== $interp round-1 kernels"; VT_NO_BLOCK_KERNEL=1 python3 tools/prof_case.py --size 512 --interp $interp --general --iters 20
  echo "== $interp lane-block kernel"; VT_BLOCK_LINEAR=1 python3 tools/prof_case.py --size 512 --interp $interp --general --iters 20
  echo "== $interp lane-block kernel, whole boxes staged"; VT_BLOCK_LINEAR=1 VT_BLOCK_NO_TRIM=1 python3 tools/prof_case.py --size 512 --interp $interp --general --iters 20
done

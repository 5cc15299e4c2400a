#!/bin/bash
# general-rotation kernels, 512^3 (25,-40,70) sxyz: lane-block kernel (lane maps 0 / 1) against the round-1 box / packed kernels
set -e
for interp in linear bspline; do
  echo "== $interp old kernels"; VT_NO_BLOCK_KERNEL=1 python3 tools/prof_case.py --size 512 --interp $interp --general --iters 20
  for lm in 0 1; do
    echo "== $interp block lm=$lm"; VT_BLOCK_LM=$lm python3 tools/prof_case.py --size 512 --interp $interp --general --iters 20
  done
done

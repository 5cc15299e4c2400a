#!/bin/bash
for a in 0 10 45; do for d in 8 16; do
  echo -n "angle=$a VT_DCH=$d : "
  VT_DCH=$d python3 tools/prof_case.py --size 1024 --interp linear --angle $a --iters 5 2>&1 | grep -v amdgpu.ids | cut -c50-130
done; done

#!/bin/bash
for interp in linear filt_bspline; do
  for tile in 2 1; do
   for dch in 8 16 32 64 128; do
    for ang in 0 45; do
      VT_DCH=$dch VT_TILE=$tile python3 tools/prof_case.py --size 512 --interp $interp --angle $ang --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg=$tile dch=$dch /"
    done
   done
  done
done
for dch in 16 32 64; do
VT_DCH=$dch VT_TILE=2 python3 tools/prof_case.py --size 1024 --interp linear --angle 45 --iters 10 2>&1 | grep -v amdgpu.ids | sed "s/^/dch=$dch /"
VT_DCH=$dch VT_TILE=2 python3 tools/prof_case.py --size 1024 --interp filt_bspline --angle 45 --iters 10 2>&1 | grep -v amdgpu.ids | sed "s/^/dch=$dch /"
done

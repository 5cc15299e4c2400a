#!/bin/bash
for tile in 0 1 5; do
for ang in 0 45 90; do
  VT_TILE=$tile python3 tools/prof_case.py --size 512 --interp linear --angle $ang --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/lin cfg=$tile /"
done
VT_TILE=$tile python3 tools/prof_case.py --size 1024 --interp linear --angle 45 --iters 10 2>&1 | grep -v amdgpu.ids | sed "s/^/lin cfg=$tile /"
done
for tile in 0 1 3; do
for ang in 0 45 90; do
  VT_TILE=$tile python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $ang --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/cub cfg=$tile /"
done
VT_TILE=$tile python3 tools/prof_case.py --size 1024 --interp filt_bspline --angle 45 --iters 10 2>&1 | grep -v amdgpu.ids | sed "s/^/cub cfg=$tile /"
done

#!/bin/bash
# round-aware chunk counts (default) vs the fixed chunk depth they replace (VT_DCH)
run() { python3 tools/prof_case.py "$@" 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-150; }
for size in 512 256 384 640; do
  for a in 0 30 45; do
    echo -n "cubic $size a=$a default : "; run --size $size --interp filt_bspline --angle $a --iters 20
    echo -n "cubic $size a=$a DCH=64  : "; VT_DCH=64 run --size $size --interp filt_bspline --angle $a --iters 20
    echo -n "linear $size a=$a default : "; run --size $size --interp linear --angle $a --iters 20
    echo -n "linear $size a=$a DCH=16  : "; VT_DCH=16 run --size $size --interp linear --angle $a --iters 20
  done
done

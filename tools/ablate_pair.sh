#!/bin/bash
# Ablation of the pair marching kernel (library built with `make EXTRA=-DVT_EXPERIMENTS`): which part of the step costs what.
run() { python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $1 --iters 20 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-45; }
for a in 0 30; do
  echo -n "a=$a full            : "; run $a
  echo -n "a=$a no stores       : "; VT_EXP_NOSTORE=1 run $a
  echo -n "a=$a no loads        : "; VT_EXP_NOLOAD=1 run $a
  echo -n "a=$a no LDS reads    : "; VT_EXP_NOLDS=1 run $a
  echo -n "a=$a no loads+stores : "; VT_EXP_NOSTORE=1 VT_EXP_NOLOAD=1 run $a
  echo -n "a=$a no ld+st+LDS    : "; VT_EXP_NOSTORE=1 VT_EXP_NOLOAD=1 VT_EXP_NOLDS=1 run $a
  echo -n "a=$a no loads+LDS    : "; VT_EXP_NOLOAD=1 VT_EXP_NOLDS=1 run $a
done

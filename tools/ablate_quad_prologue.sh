#!/bin/bash
# how much of affine_march4 is the per-workgroup set-up: the marching loop skipped (VT_EXP_NOLOOP, lib_b = -DVT_EXPERIMENTS build), 1024^3
export VT_LIB=$(pwd)/voltools_amd/lib_b/libvoltools_hip.so
for interp in filt_bspline linear; do
  for v in "" "VT_EXP_NOLOOP=1" "VT_EXP_NOLOOP=1 VT_QUAD_ROWS=-1" "VT_EXP_NOLOOP=1 VT_EXP_NOLDS=1"; do   # the last one: empty workgroups
    echo "== $interp [$v]"; env $v python3 tools/prof_case.py --size 1024 --interp $interp --angle 30 --iters 20 | cut -c1-130
  done
done

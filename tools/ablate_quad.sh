#!/bin/bash
# Ablation of the plane-quad marching kernel: which part of a step costs what.  Needs the experiment build:
#   make -C voltools_amd/csrc OUTDIR=../lib_b EXTRA=-DVT_EXPERIMENTS     (VT_LIB points the loader at it)
# usage: tools/ablate_quad.sh [size] [libdir]      -- use size 1024: below ~0.1 ms per launch the Python call rate, not the GPU, is timed
size=${1:-1024}
export VT_LIB=$(pwd)/voltools_amd/${2:-lib_b}/libvoltools_hip.so
run() { python3 tools/prof_case.py --size $size --interp $2 --angle $1 --iters 20 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-48; }
for interp in linear filt_bspline; do
for a in 0 30; do
  echo -n "$interp a=$a full            : "; run $a $interp
  echo -n "$interp a=$a no stores       : "; VT_EXP_NOSTORE=1 run $a $interp
  echo -n "$interp a=$a no loads        : "; VT_EXP_NOLOAD=1 run $a $interp
  echo -n "$interp a=$a no loads+stores : "; VT_EXP_NOSTORE=1 VT_EXP_NOLOAD=1 run $a $interp
  echo -n "$interp a=$a no ld+st+LDS    : "; VT_EXP_NOSTORE=1 VT_EXP_NOLOAD=1 VT_EXP_NOLDS=1 run $a $interp
  echo -n "$interp a=$a set-up only     : "; VT_EXP_NOLOOP=1 run $a $interp
done
done

#!/bin/bash
# ablation of the packed-footprint kernel (lib_b = make OUTDIR=../lib_b EXTRA=-DVT_EXPERIMENTS): 512^3 trilinear general rotation
export VT_LIB=$(pwd)/voltools_amd/lib_b/libvoltools_hip.so
for v in "" "VT_EXP_NOSTORE=1" "VT_EXP_NOLOAD=1" "VT_EXP_NOLDS=1" "VT_EXP_NOLOAD=1 VT_EXP_NOSTORE=1" "VT_EXP_NOLOAD=1 VT_EXP_NOLDS=1" "VT_EXP_NOLOAD=1 VT_EXP_NOLDS=1 VT_EXP_NOSTORE=1"; do
  echo "== linear [$v]"; env $v python3 tools/prof_case.py --size 512 --interp linear --general --iters 20 | cut -c1-110
done

#!/bin/bash
# Row kernel against the exchange path (VT_ROWS=0) for axis-2 rotations with integer / fractional offsets along w, one process per case
for ip in filt_bspline linear; do for a in 33 80; do for sh in 0 0.5 2; do for rows in 1 0; do
  echo -n "VT_ROWS=$rows shift $sh "; VT_ROWS=$rows python3 tools/prof_case.py --size 512 --interp $ip --axis2 --angle $a --shift2 $sh --iters 50 2>&1 | grep "ms/launch" | awk '{print $1,$2,$3,$4,$5,$6,$7}'
done; done; done; done

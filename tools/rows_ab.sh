#!/bin/bash
# Row kernel (kind 10) A/B, one variant per process: two row buffers (default) / one run per workgroup (VT_ROWS_DB=0) / two buffers on 4 x 8 tiles (2)
for a in 33 80; do
  for ip in linear filt_bspline; do
    for v in 1 0 2; do
      echo -n "VT_ROWS_DB=$v "; VT_ROWS_DB=$v python3 tools/prof_case.py --size ${1:-512} --interp $ip --axis2 --angle $a --iters 50 2>&1 | grep "ms/launch" | awk '{print $1,$2,$3,$6,$7,$13,$14,$15}'
    done
  done
done

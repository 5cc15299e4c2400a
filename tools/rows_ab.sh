#!/bin/bash
# Row kernel (kind 10) A/B, one variant per process.  usage: tools/rows_ab.sh <size> VAR=VAL ... ("-" = default)
size=${1:-512}; shift
for rep in 1 2; do
for a in 33 80; do
  for ip in linear filt_bspline; do
    for v in "$@"; do
      if [ "$v" = "-" ]; then echo -n "default "; python3 tools/prof_case.py --size $size --interp $ip --axis2 --angle $a --iters 50 2>&1 | grep "ms/launch" | awk '{print $1,$2,$3,$6,$7,$13,$14,$15}'
      else echo -n "$v "; env $v python3 tools/prof_case.py --size $size --interp $ip --axis2 --angle $a --iters 50 2>&1 | grep "ms/launch" | awk '{print $1,$2,$3,$6,$7,$13,$14,$15}'; fi
    done
  done
done
done

#!/bin/bash
# blocked tile orders of the pair kernel at 512^3: time (20 launches) and HBM reads   usage: tools/blk_ab.sh <angle> "bh bw" ...
a=$1; shift
for b in "$@"; do set -- $b
  echo -n "a=$a BLK=$1x$2 : "
  VT_BLK_H=$1 VT_BLK_W=$2 python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $a --iters 20 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-45
  VT_BLK_H=$1 VT_BLK_W=$2 bash tools/fetch_case.sh b${a}_$1_$2 --size 512 --interp filt_bspline --angle $a | head -1 | sed 's/.*read MB/    read MB/'
done

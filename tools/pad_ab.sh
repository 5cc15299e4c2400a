#!/bin/bash
for a in 3 9 15 144 171 177; do for pad in default 0 4 8; do
  echo -n "angle=$a pad=$pad : "
  if [ $pad = default ]; then python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $a --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-140
  else VT_LXPAD=$pad python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $a --iters 10 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-140; fi
done; done

#!/bin/bash
# Does the pair kernel's run time follow whole "rounds" of resident workgroups?  512^3: 512 in-plane tiles x nchunks
# workgroups, 768 resident (3 per CU): chunk depths that give 6.0 / 4.0 / 2.0 rounds vs the default 64 (5.33 rounds).
for a in 0 30; do
for d in 52 58 64 74 86 104 128 172; do
  echo -n "angle=$a VT_DCH=$d : "
  VT_DCH=$d python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $a --iters 20 2>&1 | grep -v amdgpu.ids | sed 's/.*kernel=/kernel=/' | cut -c1-60
done; done

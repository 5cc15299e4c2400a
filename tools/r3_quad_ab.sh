#!/bin/bash
# Round-3 A/B runs of the plane-quad marching kernel (one process per size and interpolation, variants interleaved; tools/march_ab.py).
#   tools/r3_quad_ab.sh zid      integer-offset trilinear kernel on / off, its chunk depth, lane mapping          (profiles/r03_ab_*.txt)
#   tools/r3_quad_ab.sh depth    finer chunk depths, tile shapes of the cubic kernel per angle
#   tools/r3_quad_ab.sh rows     cubic kernel per angle: column-aligned rows (S = 0) vs the 16-candidate model vs packed rows
case ${1:-zid} in
zid)
  python3 tools/march_ab.py --size 512 --interp linear --flags 0 --angles 0 180 3 --rounds 3 --env "" VT_QUAD_PERM=0 VT_QUAD_ZID=0 "VT_QUAD_ZID=0,VT_QUAD_PERM=0" VT_ZID_DCH=16 VT_ZID_DCH=32 VT_ZID_DCH=48
  python3 tools/march_ab.py --size 512 --interp filt_bspline --flags 0 --angles 0 180 3 --rounds 3 --env "" VT_QUAD_PERM=0
  python3 tools/march_ab.py --size 1024 --interp linear --flags 0 --angles 0 180 6 --rounds 2 --env "" VT_QUAD_PERM=0 VT_QUAD_ZID=0 VT_ZID_DCH=16 VT_ZID_DCH=24 VT_ZID_DCH=48 VT_ZID_DCH=64
  python3 tools/march_ab.py --size 1024 --interp filt_bspline --flags 0 --angles 0 180 6 --rounds 2 --env "" VT_QUAD_PERM=0 ;;
depth)
  python3 tools/march_ab.py --size 1024 --interp linear --flags 0 --angles 0 180 6 --rounds 2 --env "" VT_ZID_DCH=8 VT_ZID_DCH=12 VT_ZID_DCH=20 "VT_ZID_DCH=16,VT_TILE=2" "VT_ZID_DCH=16,VT_TILE=3"
  python3 tools/march_ab.py --size 512 --interp linear --flags 0 --angles 0 180 3 --rounds 3 --env "" VT_ZID_DCH=20 VT_ZID_DCH=28 VT_ZID_DCH=12 "VT_TILE=2" "VT_TILE=3"
  python3 tools/march_ab.py --size 512 --interp filt_bspline --flags 0 --angles 0 48 3 --rounds 3 --per-angle --env "" VT_TILE=2 VT_TILE=4 VT_TILE=3 VT_DCH=32 VT_DCH=128 VT_QUAD_ROWS=-1 ;;
rows)
  python3 tools/march_ab.py --size 512 --interp filt_bspline --flags 0 --angles 0 91 1.5 --rounds 3 --per-angle --env "" VT_QUAD_ROWS=-1 VT_QUAD_PERM=0 VT_QUAD_ROWS=0 ;;
esac

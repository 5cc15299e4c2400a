#!/bin/bash
# round 3, A/B 2: chunk depth of the integer-offset trilinear kernel (finer), tile shapes of the cubic kernel per angle
python3 tools/march_ab.py --size 1024 --interp linear --flags 0 --angles 0 180 6 --rounds 2 --env "" VT_ZID_DCH=8 VT_ZID_DCH=12 VT_ZID_DCH=20 "VT_ZID_DCH=16,VT_TILE=2" "VT_ZID_DCH=16,VT_TILE=3"
python3 tools/march_ab.py --size 512 --interp linear --flags 0 --angles 0 180 3 --rounds 3 --env "" VT_ZID_DCH=20 VT_ZID_DCH=28 VT_ZID_DCH=12 "VT_TILE=2" "VT_TILE=3"
python3 tools/march_ab.py --size 512 --interp filt_bspline --flags 0 --angles 0 48 3 --rounds 3 --per-angle --env "" VT_TILE=2 VT_TILE=4 VT_TILE=3 VT_DCH=32 VT_DCH=128 VT_QUAD_ROWS=-1

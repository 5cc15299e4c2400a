#!/bin/bash
# round 3, A/B 1: lane permutation and integer-offset trilinear kernel of affine_march4 (one process per size, variants interleaved)
python3 tools/march_ab.py --size 512 --interp linear --flags 0 --angles 0 180 3 --rounds 3 --env "" VT_QUAD_PERM=0 VT_QUAD_ZID=0 "VT_QUAD_ZID=0,VT_QUAD_PERM=0" VT_ZID_DCH=16 VT_ZID_DCH=32 VT_ZID_DCH=48
python3 tools/march_ab.py --size 512 --interp filt_bspline --flags 0 --angles 0 180 3 --rounds 3 --env "" VT_QUAD_PERM=0
python3 tools/march_ab.py --size 1024 --interp linear --flags 0 --angles 0 180 6 --rounds 2 --env "" VT_QUAD_PERM=0 VT_QUAD_ZID=0 VT_ZID_DCH=16 VT_ZID_DCH=24 VT_ZID_DCH=48 VT_ZID_DCH=64
python3 tools/march_ab.py --size 1024 --interp filt_bspline --flags 0 --angles 0 180 6 --rounds 2 --env "" VT_QUAD_PERM=0

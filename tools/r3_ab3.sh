#!/bin/bash
# round 3, A/B 3: cubic plane-quad kernel per angle: bank-aware rows on/off, lane permutation on/off, forced S = 0
python3 tools/march_ab.py --size 512 --interp filt_bspline --flags 0 --angles 0 91 1.5 --rounds 3 --per-angle --env "" VT_QUAD_ROWS=-1 VT_QUAD_PERM=0 VT_QUAD_ROWS=0

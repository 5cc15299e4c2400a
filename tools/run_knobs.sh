python tools/march_ab.py --size 1024 --flags 0 --env "" VT_QUAD_NT=0 VT_TILE=4 "VT_TILE=4,VT_QUAD_NT=0" --angles 0 90 15 --rounds 3 --reps 2
python tools/march_ab.py --size 512 --flags 0 --env "" VT_QUAD_NT=0 VT_TILE=4 "VT_TILE=4,VT_QUAD_NT=0" "VT_DCH=24" "VT_DCH=24,VT_QUAD_NT=0" --angles 0 90 15 --rounds 4

#!/bin/bash
python3 tools/angle_sweep.py --interp filt_bspline 2>&1 | grep -v amdgpu
python3 tools/angle_sweep.py --interp filt_bspline --size 1024 --step 15 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline 2>/dev/null | tail -1

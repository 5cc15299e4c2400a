#!/bin/bash
for t in 0 3 4; do
for a in 0 45; do VT_TILE=$t python3 tools/prof_case.py --size 512 --interp linear --angle $a --iters 20 2>&1 | grep -v amdgpu | cut -c1-150 | sed "s/^/cfg=$t /"; done
VT_TILE=$t python3 tools/prof_case.py --size 1024 --interp linear --angle 45 --iters 10 2>&1 | grep -v amdgpu | cut -c1-150 | sed "s/^/cfg=$t /"
done

#!/bin/bash
# HBM read bytes (FETCH_SIZE, x2-corrected per MI355X_MICROARCH.md) of the marching kernel for one prof_case configuration:
#   tools/fetch_case.sh <tag> <prof_case args...>      (environment overrides such as VT_DCH pass through)
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/fetch_$tag
rm -rf $out
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/prof_case.py "$@" --iters 4 > $out.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'affine_march' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
if agg['FETCH_SIZE']:
    print(sys.argv[1], 'read MB (x2 corrected):', round(2 * 1024 * sum(agg['FETCH_SIZE']) / len(agg['FETCH_SIZE']) / 1e6), 'launches', len(agg['FETCH_SIZE']))
PY
grep "ms/launch" $out.log | sed 's/.*kernel=/kernel=/' | cut -c1-110

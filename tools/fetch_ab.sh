#!/bin/bash
# HBM read bytes of the pair kernel for two builds: tools/fetch_ab.sh <angle>
export TMPDIR=/tmp
a=${1:-20}
for L in voltools_amd/lib/libvoltools_hip.so voltools_amd/lib_b/libvoltools_hip.so; do
  out=gpurun_out/fetch_$(basename $(dirname $L))
  rm -rf $out
  VT_LIB=$L timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/prof_case.py --size 512 --interp filt_bspline --angle $a --iters 5 > $out.log 2>&1
  python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'zpair<' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
print(sys.argv[1], {k: round(sum(v) / len(v)) for k, v in agg.items()}, 'read MB (x2 corrected):', round(2 * 1024 * sum(agg['FETCH_SIZE']) / len(agg['FETCH_SIZE']) / 1e6))
PY
  grep "ms/launch" $out.log | cut -c1-120
done

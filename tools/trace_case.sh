#!/bin/bash
# Kernel-trace durations of one prof_case.py configuration: tools/trace_case.sh <tag> <prof_case args...>  (environment switches are inherited)
export TMPDIR=/tmp
tag=$1; shift
out=$(pwd)/gpurun_out/trace_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/prof_case.py "$@" > $out/stdout.txt 2> $out/log.txt
f=$(ls $out/*/*kernel_stats.csv | head -1)
echo "== $tag: $(grep -v amdgpu $out/stdout.txt | cut -c1-90)"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'affine' in r['Name'] or 'relayout' in r['Name']:
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us min {float(r['MinNs'])/1e3:9.2f} max {float(r['MaxNs'])/1e3:9.2f}")
PY

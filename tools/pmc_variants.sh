#!/bin/bash
# One PMC pass per variant, counters of the affine kernels averaged per launch:
#   tools/pmc_variants.sh "<counters>" "<prof_case args>" "VAR=val,VAR2=val" "" "VAR=val" ...
# Environment variants are exported before rocprofv3 starts (the program after `--` is python itself: no env / shell hop).
export TMPDIR=/tmp
counters=$1; shift
args=$1; shift
out=$(pwd)/gpurun_out/pmc_variants; mkdir -p $out
for variant in "$@"; do
  (
    IFS=',' read -ra kv <<< "$variant"
    for a in "${kv[@]}"; do [ -n "$a" ] && export "$a"; done
    d=$out/run_$$_$RANDOM; mkdir -p $d
    rocprofv3 --pmc $counters --output-format csv -d $d -- python3 tools/prof_case.py $args > $d/stdout.txt 2> $d/log.txt
    python3 - "$d" "$variant" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'affine' in r['Kernel_Name'] or 'prefilter' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
line = open(sys.argv[1] + '/stdout.txt').read().strip().splitlines()[-1:] 
for k, c in acc.items():
    print(f"[{sys.argv[2] or 'default'}] {k}: " + '  '.join(f"{n}={sum(v)/len(v):.1f} (x{len(v)})" for n, v in sorted(c.items())))
print('    ', (line[0] if line else '')[:200])
PY
    rm -rf $d
  )
done

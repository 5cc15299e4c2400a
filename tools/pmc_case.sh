#!/bin/bash
# PMC instruction / wait mix of one prof_case.py configuration, per 64 voxels: tools/pmc_case.sh <kernel substring> <prof_case args...>
# (environment switches such as VT_LIB / VT_EXP_* are inherited)
export TMPDIR=/tmp
kern=$1; shift
out=$(pwd)/gpurun_out/pmc_case; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/a -- python3 tools/prof_case.py "$@" > $out/stdout.txt 2> $out/log.txt
python3 - "$out/a" "$kern" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
g = 512 ** 3 / 64.0
print({k: round(sum(v) / len(v) / g, 1) for k, v in acc.items()}, '(per 64 voxels of a 512^3 launch)')
PY

#!/bin/bash
# instruction mix of the affine_march4 set-up alone (marching loop skipped): PMC pass over the VT_EXP_NOLOOP run, 1024^3
export TMPDIR=/tmp
export VT_LIB=$(pwd)/voltools_amd/lib_b/libvoltools_hip.so
export VT_EXP_NOLOOP=1
out=$(pwd)/gpurun_out/pmc_prologue; rm -rf $out; mkdir -p $out
for interp in linear filt_bspline; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/$interp -- python3 tools/prof_case.py --size 1024 --interp $interp --angle 30 --iters 5 > /dev/null 2> $out/$interp.log
  python3 - "$out/$interp" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'affine_march4' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
w = sum(acc['SQ_WAVES']) / max(1, len(acc['SQ_WAVES']))
print(sys.argv[1].split('/')[-1], 'waves per launch', w, {k: round(sum(v) / len(v) / w, 1) for k, v in acc.items() if k != 'SQ_WAVES'}, '(per wave)')
PY
done

#!/bin/bash
# usage: tools/pmc_small.sh <tag> <prof_case args...>  -- two SQ passes only
tag=$1; shift
export TMPDIR=/tmp
out=$(pwd)/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/p0 -- python3 tools/prof_case.py "$@" > $out/p0.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $out/p1 -- python3 tools/prof_case.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_IFETCH SQ_INSTS_WAVE32_LDS --output-format csv -d $out/p2 -- python3 tools/prof_case.py "$@" > $out/p2.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'][:50]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'affine' not in k: continue
    print('KERNEL', k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} mean={sum(v)/len(v):.6g}')
PY
tail -2 $out/p1.log | head -1

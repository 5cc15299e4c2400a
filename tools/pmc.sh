#!/bin/bash
# usage: tools/pmc.sh <tag> <prof_case args...>   -- PMC passes (each its own run; never combined with tracing)
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
)
i=0
for pass in "${passes[@]}"; do
  rocprofv3 --pmc $pass --output-format csv -d $out/p$i -- python3 tools/prof_case.py "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    print('KERNEL', k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}')
PY

#!/bin/bash
# Round profile of the bench command (run on the GPU box via gpurun):
#   1. rocprofv3 --kernel-trace --stats  -> per-kernel average durations (must agree with bench.py's HIP-event timing)
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE ...) -> HBM-side traffic per launch of the dominant kernel
# Outputs under gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
tag=${1:-r01}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/profiles_$tag
rm -rf $out; mkdir -p $out
args="--steps 180 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/bench_under_trace.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -- python3 bench.py $args > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 bench.py $args > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/pmc_sq -- python3 bench.py $args > /dev/null 2> $out/pmc_sq.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc_lds -- python3 bench.py $args > /dev/null 2> $out/pmc_lds.log
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
# kernel stats
stats = {}
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        stats[row['Name']] = {k: row[k] for k in row if k != 'Name'}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
summary = {'tag': tag, 'kernel_stats': stats, 'pmc_mean_per_launch': {}}
for k, d in agg.items():
    if 'affine' not in k and 'prefilter' not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    m['launches_sampled'] = max(len(v) for v in d.values())
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reports exactly half of the bytes of a wide (16 B/lane) coalesced
        # streaming read on gfx950 -> doubled; WRITE_SIZE (KiB) is exact for streaming stores
        m['hbm_read_bytes_corrected'] = 2.0 * m['FETCH_SIZE'] * 1024
        m['hbm_write_bytes'] = m['WRITE_SIZE'] * 1024
        m['hbm_traffic_bytes'] = m['hbm_read_bytes_corrected'] + m['hbm_write_bytes']
    summary['pmc_mean_per_launch'][k] = m
json.dump(summary, open(out + f'/{tag}_summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1)[:6000])
PY

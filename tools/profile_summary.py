"""Aggregate one tools/profile_round.sh case: kernel-trace stats + PMC means per kernel -> <dir>/<tag>_<name>_summary.json.
FETCH_SIZE is doubled and WRITE_SIZE taken as is (KiB), as MI355X_MICROARCH.md's HBM section prescribes for gfx950."""
import collections
import csv
import glob
import json
import sys

d, tag, name = sys.argv[1:4]
stats = {}
for f in glob.glob(d + '/trace/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        stats[row['Name']] = {k: row[k] for k in row if k != 'Name'}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/pmc*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
summary = {'tag': tag, 'case': name, 'kernel_stats': stats, 'pmc_mean_per_launch': {}}
for k, c in agg.items():
    if not any(s in k for s in ('affine', 'prefilter', 'relayout', 'transpose', 'plane_sum')):
        continue
    m = {cn: sum(v) / len(v) for cn, v in c.items()}
    m['launches_sampled'] = max(len(v) for v in c.values())
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        m['hbm_read_bytes_corrected'] = 2.0 * m['FETCH_SIZE'] * 1024
        m['hbm_write_bytes'] = m['WRITE_SIZE'] * 1024
        m['hbm_traffic_bytes'] = m['hbm_read_bytes_corrected'] + m['hbm_write_bytes']
    if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
        m['l2_hit_rate'] = m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum'])
    summary['pmc_mean_per_launch'][k] = m
json.dump(summary, open(f'{d}/{tag}_{name}_summary.json', 'w'), indent=1)
for k, s in sorted(stats.items(), key=lambda kv: -float(kv[1].get('TotalDurationNs', 0)))[:6]:
    m = summary['pmc_mean_per_launch'].get(k, {})
    print(f"{k[:90]}\n    calls {s.get('Calls')} avg {float(s.get('AverageNs', 0)) / 1e3:.1f} us  "
          f"traffic {m.get('hbm_traffic_bytes', 0) / 1e9:.3f} GB  L2 hit {m.get('l2_hit_rate', 0):.2f}  "
          f"LDS conflict/active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_LDS_IDX_ACTIVE', 1)):.2f}  "
          f"SALU/VALU {m.get('SQ_INSTS_SALU', 0) / max(1.0, m.get('SQ_INSTS_VALU', 1)):.2f}")

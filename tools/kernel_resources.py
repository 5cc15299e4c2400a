#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of one .hip file (device-only compile to assembly, metadata parsed).
usage: tools/kernel_resources.py voltools_amd/csrc/vt_kernels_quad.hip [name-filter] [-DVT_LEGACY ...]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
extra = [a for a in sys.argv[2:] if a.startswith('-')]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, 'k.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=fast', '--cuda-device-only', '-S',
                    '-w', src, '-o', out] + extra, check=True)
    txt = open(out).read()
meta = txt[txt.index('amdhsa.kernels:'):]
for blk in meta.split('  - .agpr_count:')[1:]:
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void vt::', '')
    if filt and filt not in name:
        continue
    print(f"{name:58s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')} "
          f"scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size')}")

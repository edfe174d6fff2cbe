#!/usr/bin/env python
"""Kernel resource table of one HIP source: VGPRs / AGPRs / spills / scratch / occupancy per kernel, from
hipcc -Rpass-analysis=kernel-resource-usage.   python scripts/kres.py auto-dynamic-deeplab_amd/csrc/conv3.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'include')
r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I' + inc, '-ffp-contract=off',
                    '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r'remark: +(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(anonymous namespace\)::', '', d).split('(')[0]}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
print('%-60s %5s %5s %6s %7s %4s %6s' % ('kernel', 'VGPR', 'AGPR', 'spill', 'scratch', 'occ', 'LDS'))
for d in rows:
    if flt in d['name']:
        print('%-60s %5s %5s %6s %7s %4s %6s' % (d['name'][:60], d.get('VGPRs'), d.get('AGPRs'), d.get('VGPRs Spill'), d.get('ScratchSize [bytes/lane]'),
                                             d.get('Occupancy [waves/SIMD]'), d.get('LDS Size [bytes/block]')))

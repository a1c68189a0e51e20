#!/usr/bin/env python
"""hipcc -Rpass-analysis=kernel-resource-usage summary for one .hip file (VGPR/AGPR/spill/LDS/occupancy)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
                    '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.split('\n'):
    if 'error' in line:
        print(line)
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()[:150]}
        rows.append(cur)
        continue
    for key in ('VGPRs', 'AGPRs', 'VGPRs Spill', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs'):
        m = re.search(r'remark:\s+%s: (\d+)' % re.escape(key), line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for c in rows:
    if flt and flt not in c['name']:
        continue
    print('%-92s v=%-4s a=%-4s spill=%-4s scratch=%-5s occ=%s' % (
        c['name'], c.get('VGPRs'), c.get('AGPRs'), c.get('VGPRs Spill'), c.get('ScratchSize [bytes/lane]'),
        c.get('Occupancy [waves/SIMD]')))

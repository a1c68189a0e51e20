"""gpurun_out/r05_final (scripts/r05_final.sh on the GPU box) -> the tracked evidence under profiles/ and the HBM traffic
table bench.py reads (profiles/pmc_hbm_traffic.json).  usage: python scripts/r05_collect.py [gpurun_out/r05_final]"""
import json
import os
import re
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r05_final'
P = 'profiles'


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(P, b))
        print('copied', b)


cp('bench20_full.json', 'r05_bench_steps20_warmup5.json')
cp('bench_default_full.json', 'r05_bench_default.json')
cp('stats.md', 'r05_kernel_trace.md')
cp('timeline.md', 'r05_step_timeline.md')
cp('statsA.md', 'r05_cfgA_kernel_trace.md')
cp('gpu_tests.txt', 'r05_gpu_tests.txt')
cp('cfgC/stats.md', 'r05_cfgC_kernel_trace.md')
cp('cfgD/stats.md', 'r05_cfgD_kernel_trace.md')
cp('cfgC/timeline.md', 'r05_cfgC_timeline.md')
cp('cfgD/timeline.md', 'r05_cfgD_timeline.md')


def pmc(dirname):
    out = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        path = os.path.join(src, dirname, c + '.txt')
        if not os.path.exists(path):
            return None, ''
        for line in open(path):
            m = re.search(r'n=(\d+)\s+avg=([0-9.]+)\s+avg_us=([0-9.]+)', line)
            if not m:
                continue
            name = 'lstm_bwd' if 'lstm_bwd' in line else 'lstm_fwd' if 'lstm_fwd' in line else \
                   'optimizer' if 'optimizer_kernel' in line else None
            if name:
                out.setdefault(name, {})[c] = (float(m.group(2)), float(m.group(3)), int(m.group(1)))
    text = ''.join(open(os.path.join(src, dirname, c + '.txt')).read() for c in ('FETCH_SIZE', 'WRITE_SIZE'))
    return out, text


table = {}
md = ['# Round 5 -- HBM-side traffic of the recurrence kernels (rocprofv3 --pmc, separate passes)\n',
      'Commands (`scripts/r04_pmc.sh`, called by `scripts/r05_final.sh`): `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 '
      '--no-cpu-baseline --no-parity --no-cfgA --no-aux [workload flags]` and the same with `--pmc WRITE_SIZE` (one counter per '
      'pass, nothing but the kernel trace next to `--pmc`); KiB per launch averaged over the launches of the run '
      '(`scripts/rocpd_pmc.py`).  Bytes per launch = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md for wide coalesced '
      'reads, re-checked below on `optimizer_kernel<5>`: params + grads + 2 slots = 4 x 28.44 MB = 113.8 MB read, 85.3 MB '
      'written) + WRITE_SIZE, counters in KiB.\n']
for key, d, flags in (('5x256_bf16_B16_T778', 'pmc', '(headline)'),
                      ('2x128_f32_B16_T778', 'pmcA', '--units 128 --layers 2 --dtype f32 --classes 39 --keep-prob 0.5')):
    o, text = pmc(d)
    if not o:
        continue
    md.append('\n## %s  %s\n' % (key, flags))
    md.append('| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes per launch (2 F + W) | avg us (counter pass) |\n|---|---|---|---|---|\n')
    row = {}
    for k in ('lstm_fwd', 'lstm_bwd', 'optimizer'):
        if k in o and len(o[k]) == 2:
            f, w = o[k]['FETCH_SIZE'][0], o[k]['WRITE_SIZE'][0]
            b = int(round((2 * f + w) * 1024))
            md.append('| %s | %.1f | %.1f | %d | %.1f |\n' % (k, f, w, b, o[k]['FETCH_SIZE'][1]))
            if k != 'optimizer':
                row[k] = b
    table[key] = row
    md.append('\n```\n' + '\n'.join(l[:170] for l in text.splitlines()[:14]) + '\n```\n')
open(os.path.join(P, 'r05_pmc_hbm.md'), 'w').write(''.join(md))
tt = {'source': 'profiles/r05_pmc_hbm.md: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) of '
                '`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-cfgA --no-aux [workload flags]` '
                '(scripts/r04_pmc.sh); bytes per launch = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md, confirmed on '
                'optimizer_kernel: 2 x 55 570 KiB = the 113.8 MB it reads) + WRITE_SIZE, counters in KiB',
      'workloads': table}
old = json.load(open(os.path.join(P, 'pmc_hbm_traffic.json')))
for k, v in old.get('workloads', {}).items():
    table.setdefault(k, v)          # 5x512 / 5x320: round 4 / round 3 rows (kernels unchanged since)
tt['source'] += '; the 5x512 row is round 4\'s (profiles/r04_pmc_hbm.md), the 5x320 row round 3\'s (profiles/r03_pmc_hbm.md): those kernels are unchanged'
json.dump(tt, open(os.path.join(P, 'pmc_hbm_traffic.json'), 'w'), indent=1)
print(json.dumps(table, indent=1))

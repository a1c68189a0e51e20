"""gpurun_out/r03_final (scripts/r03_final.sh on the GPU box) -> the tracked evidence under profiles/ and the HBM traffic
table bench.py reads (profiles/pmc_hbm_traffic.json).  usage: python scripts/r03_collect.py [gpurun_out/r03_final]"""
import json
import os
import re
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r03_final'
P = 'profiles'


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(P, b))
        print('copied', b)


cp('bench20.json', 'r03_bench_steps20_warmup5.json')
cp('bench_default.json', 'r03_bench_default.json')
cp('trace/stats.md', 'r03_kernel_trace.md')
cp('trace/timeline.md', 'r03_step_timeline.md')
cp('cfgC/stats.md', 'r03_cfgC_kernel_trace.md')
cp('cfgD/stats.md', 'r03_cfgD_kernel_trace.md')
if os.path.exists('gpurun_out/r03_full/gpu_tests.log'):
    shutil.copyfile('gpurun_out/r03_full/gpu_tests.log', os.path.join(P, 'r03_gpu_tests.txt'))


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
md = ['# Round 3 -- HBM-side traffic of the recurrence kernels (rocprofv3 --pmc, separate passes)\n',
      'Commands (`scripts/r03_pmc.sh`): `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 '
      '--no-cpu-baseline --no-parity --no-cfgA --no-aux [workload flags]` and the same with `--pmc WRITE_SIZE` (one counter per '
      'pass, nothing but the kernel trace next to `--pmc`); KiB per launch averaged over the launches of the run '
      '(`scripts/rocpd_pmc.py`).  Bytes per launch = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md for wide coalesced '
      'reads, re-checked below on `optimizer_kernel<5>`: params + grads + 2 slots = 4 x 28.44 MB = 113.8 MB read, 85.3 MB '
      'written) + WRITE_SIZE, counters in KiB.\n']
for key, d, flags in (('5x256_bf16_B16_T778', 'pmc', '(headline)'), ('5x512_bf16_B32_T778', 'pmc512', '--units 512 --batch 32'),
                      ('2x128_f32_B16_T778', 'pmcA', '--units 128 --layers 2 --dtype f32 --classes 39 --keep-prob 0.5'),
                      ('5x320_bf16_B16_T778', 'pmc320', '--units 320')):
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
open(os.path.join(P, 'r03_pmc_hbm.md'), 'w').write(''.join(md))
tt = {'source': 'profiles/r03_pmc_hbm.md: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) of '
                '`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-cfgA --no-aux [workload flags]` '
                '(scripts/r03_pmc.sh); bytes per launch = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md, confirmed on '
                'optimizer_kernel: 2 x 55 570 KiB = the 113.8 MB it reads) + WRITE_SIZE, counters in KiB',
      'workloads': table}
json.dump(tt, open(os.path.join(P, 'pmc_hbm_traffic.json'), 'w'), indent=1)
print(json.dumps(table, indent=1))

#!/bin/bash
# round 5: the cfg C entry of bench.py itself (back-to-back steps, no host sync) with the VGG weight gradients on / off the side lane
set -u
OUT=${1:-gpurun_out/r05_cfgC_bench_ab}
mkdir -p $OUT
for A in 0 1 0 1; do
  ASR_VGG_WGRAD_SIDE=$A python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-cfgA --aux cfgC --aux-steps 10 > $OUT/o.out 2> $OUT/o.err
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.load(open('bench_full.json'))['cfgC']
print('side=$A', 'ms/step %.2f' % d['ms_per_step'], 'median %.2f min %.2f max %.2f' % (d['step_ms']['median'], d['step_ms']['min'], d['step_ms']['max']), {k: round(v['avg_us']) for k, v in d['kernels'].items()})
PY
done

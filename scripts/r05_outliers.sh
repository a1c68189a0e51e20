#!/bin/bash
# round 5: where do the occasional 20-30 ms steps of the headline bench come from?  100 timed steps, collector on / off, three times each
set -u
OUT=${1:-gpurun_out/r05_outliers}
mkdir -p $OUT
for G in 1 0 1 0 1 0; do
  ASR_BENCH_GC=$G python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/o.out 2> $OUT/o.err
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.load(open('bench_full.json'))
print('gc=$G', 'ms/step %.3f' % d['ms_per_step'], 'median %.3f min %.3f max %.3f' % (d['step_ms']['median'], d['step_ms']['min'], d['step_ms']['max']), 'slow', d['step_ms'].get('slow_steps'))
PY
done

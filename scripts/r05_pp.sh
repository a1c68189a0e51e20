#!/bin/bash
# round 5: two poll rounds in flight in the forward recurrence (ASR_LSTM_PP): headline A/B (loss and error word checked)
set -u
OUT=${1:-gpurun_out/r05_pp}
mkdir -p $OUT
for P in ${PLIST:-0 2 4 6 0 4}; do
  rm -f bench_full.json
  ASR_LSTM_PP=$P python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/o.out 2> $OUT/o.err || tail -2 $OUT/o.err | cut -c1-200
  python - <<PY | tee -a $OUT/ab.txt
import json, os
if os.path.exists('bench_full.json'):
    d = json.load(open('bench_full.json'))
    k = d['kernels']
    print('pp=$P', 'ms/step %.3f' % d['ms_per_step'], 'median %.3f' % d['step_ms']['median'], 'lstm_fwd %.1f us lstm_bwd %.1f' % (k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us']), 'loss', d['final_loss'], 'flags', d['cluster_handoff_flags'])
else:
    print('pp=$P FAILED')
PY
done

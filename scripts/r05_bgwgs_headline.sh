#!/bin/bash
# round 5: workgroups of the weight-gradient GEMMs beside the headline's BPTT kernels (ASR_BG_WGS): interference against finishing in time
set -u
OUT=${1:-gpurun_out/r05_bgwgs_headline}
mkdir -p $OUT
for W in ${WLIST:-128 16 32 64 128 32}; do
  ASR_BG_WGS=$W python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/o.out 2> $OUT/o.err
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.load(open('bench_full.json'))
k = d['kernels']
print('bg_wgs=$W', 'ms/step %.3f' % d['ms_per_step'], 'median %.3f' % d['step_ms']['median'], 'lstm_bwd avg %.1f us' % k['lstm_bwd']['avg_us'], 'lstm_fwd %.1f' % k['lstm_fwd']['avg_us'])
PY
done

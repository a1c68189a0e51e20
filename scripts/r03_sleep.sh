#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_pin4}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for F in 0 0; do
  timeout 120 python bench.py --steps 20 --warmup 5 $Q > $OUT/b256_$RANDOM.json 2>> $OUT/err.log
done
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    d = json.load(open(p)); k = d['kernels']
    print('%-20s %.3f ms/step fwd %.1f bwd %.1f handoff %s' % (p.split('/')[-1], d['ms_per_step'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags']))
PY

#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_pin512}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for F in 0 10240 20480 30720; do
  ASR_LSTM_DFLAGS=$F timeout 120 python bench.py --steps 8 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_f$F.json 2>> $OUT/err.log
done
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    d = json.load(open(p)); k = d['kernels']
    print('%-20s %.3f ms/step fwd %.1f bwd %.1f handoff %s' % (p.split('/')[-1], d['ms_per_step'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags']))
PY

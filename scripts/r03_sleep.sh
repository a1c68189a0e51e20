#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_fpin2}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "cluster or lstm" > $OUT/tests.log 2>&1
tail -2 $OUT/tests.log
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
timeout 120 python bench.py --steps 20 --warmup 5 $Q > $OUT/b256.json 2>> $OUT/err.log
for F in 0 2048; do
  ASR_LSTM_DFLAGS=$F timeout 120 python bench.py --steps 8 --warmup 3 --units 320 $Q > $OUT/b320_f$F.json 2>> $OUT/err.log
done
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    d = json.load(open(p)); k = d['kernels']
    print('%-20s %.3f ms/step fwd %.1f bwd %.1f handoff %s' % (p.split('/')[-1], d['ms_per_step'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags']))
PY

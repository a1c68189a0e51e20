#!/bin/bash
# A/B of the forward recurrence on H/32 CUs x 4 waves (ASR_LSTM_FWD_HS=32) against H/64 CUs x 8 waves (default)
set -u
OUT=${1:-gpurun_out/r03_hs}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for HSV in 64 32; do
  ASR_LSTM_FWD_HS=$HSV timeout 120 python bench.py --steps 20 --warmup 5 $Q > $OUT/b256_hs$HSV.json 2> $OUT/b256_hs$HSV.err
  ASR_LSTM_FWD_HS=$HSV timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_hs$HSV.json 2> $OUT/b512_hs$HSV.err
done
ASR_LSTM_FWD_HS=32 timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "cluster" > $OUT/tests_hs32.log 2>&1
echo "cluster tests with HS=32: $(tail -1 $OUT/tests_hs32.log)"
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    try:
        d = json.load(open(p)); k = d['kernels']
        print('%-40s %.0f frames/s %.3f ms/step (median %.3f host %.2f wait %.2f) fwd %.1f bwd %.1f us handoff %s loss %.4f' % (p.split('/')[-1], d['value'], d['ms_per_step'],
              d['step_ms']['median'], d['step_ms']['host_issue_mean'], d['step_ms']['host_wait_for_device_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags'], d['final_loss']))
    except Exception as e:
        print(p, 'ERR', e)
PY

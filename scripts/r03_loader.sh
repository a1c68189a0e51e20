#!/bin/bash
# BPTT loader wave (ASR_LSTM_LOADER): cluster parity tests, then A/B on the headline and the 5x512 B=32 step
set -u
OUT=${1:-gpurun_out/r03_loader}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "cluster or lstm" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for F in 1 0 1 0; do
  ASR_LSTM_LOADER=$F timeout 120 python bench.py --steps 30 --warmup 5 $Q > $OUT/b256_l${F}_$RANDOM.json 2>> $OUT/err.log
done
for F in 1 0; do
  ASR_LSTM_LOADER=$F timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_l$F.json 2>> $OUT/err.log
  ASR_LSTM_LOADER=$F timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 --tmax 1600 --tmin 800 $Q > $OUT/b512T_l$F.json 2>> $OUT/err.log
done
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    try:
        d = json.load(open(p)); k = d['kernels']
        print('%-24s %.0f frames/s %.3f ms/step (median %.3f host %.2f) fwd %.1f bwd %.1f us handoff %s loss %.4f' % (p.split('/')[-1], d['value'], d['ms_per_step'],
              d['step_ms']['median'], d['step_ms']['host_issue_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags'], d['final_loss']))
    except Exception as e:
        print(p, 'ERR', e)
PY
tail -5 $OUT/err.log

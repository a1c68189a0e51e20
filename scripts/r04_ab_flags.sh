#!/bin/bash
# A/B of ASR_LSTM_DFLAGS settings on the headline (5x256 B=16), 5x512 B=32 and 5x320 steps (round 4: reads the compact line)
#   usage  r04_ab_flags.sh OUT "0 1024 ..." [extra env assignments]
set -u
OUT=${1:-gpurun_out/r04_fl}
FLAGS=${2:-"0 1024"}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for F in $FLAGS; do
  ASR_LSTM_DFLAGS=$F timeout 120 python bench.py --steps 20 --warmup 5 $Q > $OUT/b256_f$F.json 2> $OUT/b256_f$F.err
  ASR_LSTM_DFLAGS=$F timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_f$F.json 2> $OUT/b512_f$F.err
  ASR_LSTM_DFLAGS=$F timeout 120 python bench.py --steps 10 --warmup 3 --units 320 $Q > $OUT/b320_f$F.json 2> $OUT/b320_f$F.err
done
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1]); k = d['kernels']
        print('%-24s %.0f frames/s %.3f ms/step (median %.3f host %.2f) fwd %.1f bwd %.1f us handoff %s loss %.4f' % (p.split('/')[-1], d['value'], d['ms_per_step'],
              d['step_ms']['median'], d['step_ms']['host_issue_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags'], d['final_loss']))
    except Exception as e:
        print(p, 'ERR', e)
PY

#!/bin/bash
# round-2 GPU call 1: headline-shape parity of the bf16 cluster kernels, EARLY variant, driver-settings bench
set -u
OUT=gpurun_out/r02_c1
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -s -k "headline or cfgB" > $OUT/parity.log 2>&1
echo "parity rc=$? $(tail -1 $OUT/parity.log)"
ASR_LSTM_DFLAGS=32 timeout 900 python -m pytest tests -m gpu -q -s -k "headline or cfgB or lstm_cluster_exchange" > $OUT/parity_early.log 2>&1
echo "parity EARLY rc=$? $(tail -1 $OUT/parity_early.log)"
for f in 0 32; do
  ASR_LSTM_DFLAGS=$f timeout 120 python bench.py --no-cpu-baseline > $OUT/bench_$f.json 2> $OUT/bench_$f.err
  ASR_LSTM_DFLAGS=$f timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench20_$f.json 2> $OUT/bench20_$f.err
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r02_c1/bench*.json')):
    try:
        d = json.load(open(p)); k = d['kernels']
        print('%s: %.0f frames/s %.3f ms/step fwd %.1f bwd %.1f ctc %.1f handoff %s' % (p, d['value'], d['ms_per_step'],
              k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], k['ctc_loss']['avg_us'], d['cluster_handoff_flags']))
    except Exception as e:
        print(p, 'ERR', e)
PY

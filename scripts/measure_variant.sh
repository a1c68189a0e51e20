#!/bin/bash
# One gpurun call that validates and measures a recurrence-kernel variant selected by ASR_LSTM_DFLAGS:
#   /usr/local/graft/bin/gpurun --timeout 300 -- 'bash scripts/measure_variant.sh 32'
# 1. the LSTM / CTC-model parity tests with the variant switched on (the flag selects kernels process-wide),
# 2. the headline bench without and with the variant (no CPU baseline leg), hand-off flags included,
# 3. the per-phase cycle counters of the forward / backward cluster kernels for both (scripts/probe_lstm_phases.py).
# Results land in gpurun_out/variant_<flags>/.
set -u
FLAGS=${1:?usage: measure_variant.sh <ASR_LSTM_DFLAGS value>}
OUT=gpurun_out/variant_${FLAGS}
mkdir -p "$OUT"
ASR_LSTM_DFLAGS=$FLAGS timeout 240 python -m pytest tests -m gpu -q -x -k "lstm or ctc_model or smoke" > "$OUT/tests.log" 2>&1
echo "tests rc=$? $(tail -1 "$OUT/tests.log")"
for f in 0 "$FLAGS"; do
  ASR_LSTM_DFLAGS=$f timeout 120 python bench.py --no-cpu-baseline > "$OUT/bench_$f.json" 2> "$OUT/bench_$f.err"
  python - "$OUT/bench_$f.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d['kernels']
print('flags %s: %.0f frames/s  %.3f ms/step  fwd %.1f us  bwd %.1f us  ctc %.1f us  handoff %s' % (
    sys.argv[1].split('_')[-1][:-5], d['value'], d['ms_per_step'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'],
    k['ctc_loss']['avg_us'], d['cluster_handoff_flags']))
PY
done

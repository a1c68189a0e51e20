#!/bin/bash
# A/B of the fp32 cluster kernels' variants on the cfg-A-shaped step: bash scripts/r02_f32_cluster_ab.sh <outdir> <dflags>...
set -u
OUT=$1; shift
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q --tb=short -m gpu \
  -k "lstm_cluster_f32 or lstm_fwd_f32 or lstm_bwd_f32 or lstm_bf16_and_wide or ctc_model_loss or handoff" > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"
grep -E "^(FAILED|ERROR)|^E  " $OUT/tests.log | head -30
for v in "$@"; do
  ASR_LSTM_DFLAGS=$v timeout 120 python bench.py --units 128 --layers 2 --dtype f32 --classes 39 --keep-prob 0.5 \
    --steps 20 --warmup 3 --no-cfgA --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  OUT=$OUT V=$v python - <<'PY'
import json, os
d = json.load(open('%s/bench_%s.json' % (os.environ['OUT'], os.environ['V']))); k = d['kernels']; p = d.get('parity') or {}
print('dflags=%s: %.0f frames/s %.3f ms/step (median %.3f) fwd %.1f bwd %.1f us handoff %s loss_rel %.2e mismatches %s' % (
    os.environ['V'], d['value'], d['ms_per_step'], d['step_ms']['median'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'],
    d['cluster_handoff_flags'], p.get('loss_rel_err_vs_oracle', -1), p.get('greedy_label_mismatch')))
PY
done

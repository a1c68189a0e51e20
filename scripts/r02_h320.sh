#!/bin/bash
# H = 320 on the five-CU cluster kernels: parity tests, then the 5x320 step with and without them
set -u
OUT=${1:-gpurun_out/r02_h320}
mkdir -p $OUT
timeout 150 python -m pytest tests/test_gpu_ops.py -q --tb=short -m gpu -k "(gradient_parity_headline_shapes and 320) or lstm_bf16_and_wide" > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"
grep -E "^(FAILED|ERROR)|^E  " $OUT/tests.log | head -30
for v in 1 0; do
  ASR_LSTM_CLUSTER_320=$v timeout 100 python bench.py --units 320 --steps 20 --warmup 3 --no-cfgA --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  OUT=$OUT V=$v python - <<'PY'
import json, os
d = json.load(open('%s/bench_%s.json' % (os.environ['OUT'], os.environ['V']))); k = d['kernels']; p = d.get('parity') or {}
print('cluster_320=%s: %.0f frames/s %.3f ms/step (median %.3f) fwd %.1f bwd %.1f us handoff %s loss_rel %.2e mismatches %s' % (
    os.environ['V'], d['value'], d['ms_per_step'], d['step_ms']['median'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'],
    d['cluster_handoff_flags'], p.get('loss_rel_err_vs_oracle', -1), p.get('greedy_label_mismatch')))
PY
done

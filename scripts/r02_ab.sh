#!/bin/bash
# A/B of an environment switch on the headline bench: bash scripts/r02_ab.sh <outdir> VAR val1 val2 ...
set -u
OUT=$1; VAR=$2; shift 2
mkdir -p $OUT
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-cfgA > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  OUT=$OUT V=$v VAR=$VAR python - <<'PY'
import json, os
d = json.load(open('%s/bench_%s.json' % (os.environ['OUT'], os.environ['V']))); k = d['kernels']
print('%s=%s: %.3f ms/step (median %.3f, host %.2f) fwd %.1f bwd %.1f ctc %.1f handoff %s' % (os.environ['VAR'], os.environ['V'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['host_issue_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], k['ctc_loss']['avg_us'], d['cluster_handoff_flags']))
PY
done

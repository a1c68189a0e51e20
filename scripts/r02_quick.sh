#!/bin/bash
# quick GPU check: a pytest selection + the 20-step bench without the CPU leg
# usage: bash scripts/r02_quick.sh <outdir> <pytest -k expression>
set -u
OUT=${1:-gpurun_out/r02_quick}
KEXPR=${2:-ctc}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "$KEXPR" > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"
grep -E "^E |Error" $OUT/tests.log | head -20
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfgA > $OUT/bench20.json 2> $OUT/bench20.err
OUT=$OUT python - <<'PY'
import json, os
d = json.load(open(os.environ['OUT'] + '/bench20.json')); k = d['kernels']
print('%.0f frames/s %.3f ms/step (median %.3f, host issue %.2f) fwd %.1f bwd %.1f ctc %.1f handoff %s' % (d['value'], d['ms_per_step'],
      d['step_ms']['median'], d['step_ms']['host_issue_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], k['ctc_loss']['avg_us'], d['cluster_handoff_flags']))
print('parity', d.get('parity'))
PY

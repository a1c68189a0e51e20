#!/bin/bash
# full GPU suite, driver-settings bench (20 steps / 5 warm-up), default bench, kernel trace + one-step timeline
# usage: bash scripts/r02_gpu_suite_bench_trace.sh [outdir under gpurun_out/]
set -u
OUT=${1:-gpurun_out/r02_c5}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc=$? $(tail -1 $OUT/gpu_tests.log)"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err
echo "bench20 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-cfgA > $OUT/bench100.json 2> $OUT/bench100.err
echo "bench100 rc=$?"
OUT=$OUT python - <<'PY'
import json, glob, os
for p in sorted(glob.glob(os.environ['OUT'] + '/bench*.json')):
    try:
        d = json.load(open(p)); k = d['kernels']
        print('%s: %.0f frames/s %.3f ms/step (median %.3f, host issue %.2f) fwd %.1f bwd %.1f ctc %.1f handoff %s' % (p, d['value'], d['ms_per_step'],
              d['step_ms']['median'], d['step_ms']['host_issue_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], k['ctc_loss']['avg_us'], d['cluster_handoff_flags']))
        if d.get('parity'): print('   parity', d['parity'])
        if d.get('h2d_inclusive'): print('   h2d', d['h2d_inclusive']['ms_per_step'])
        if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['seconds_per_step'])
        if d.get('cfgA'): print('   cfgA', d['cfgA']['value'], d['cfgA']['ms_per_step'], d['cfgA']['parity'], d['cfgA']['kernels'])
    except Exception as e:
        print(p, 'ERR', e)
PY
bash scripts/r02_trace.sh $OUT/trace

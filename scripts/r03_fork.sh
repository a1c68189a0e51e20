#!/bin/bash
# A/B of the main-stream marker count (ASR_FORK_ONCE) on the headline and the 5x512 B=32 step + one-step timeline with
# the new default; library-GEMM probe.   usage: r03_fork.sh [OUT]
set -u
OUT=${1:-gpurun_out/r03_fork}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for F in 1 0 1 0; do
  for i in a; do
  ASR_FORK_ONCE=$F timeout 120 python bench.py --steps 30 --warmup 5 $Q > $OUT/b256_f${F}_$RANDOM.json 2>> $OUT/err.log
  done
done
ASR_FORK_ONCE=1 timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_f1.json 2>> $OUT/err.log
ASR_FORK_ONCE=0 timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_f0.json 2>> $OUT/err.log
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
bash scripts/r03_trace.sh $OUT/trace > $OUT/trace.log 2>&1
head -3 $OUT/trace/timeline.md
timeout 300 python scripts/probe_matmul.py 2>&1 | tee $OUT/matmul.txt

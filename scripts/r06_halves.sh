#!/bin/bash
# round 6: the encoder's two half-batch pipelines (ASR_ENC_HALVES): parity tests that run them (B >= 17), then cfg C / D / E
# of the bench with and without
set -u
OUT=${1:-gpurun_out/r06_halves}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "cfgC or two_pipelines" > $OUT/tests_cfg.txt 2>&1 ) 2> $OUT/tests_cfg.time
tail -3 $OUT/tests_cfg.txt | cut -c1-300; grep real $OUT/tests_cfg.time
for hv in 0 1; do
  ASR_ENC_HALVES=$hv timeout 600 python bench.py --steps 5 --warmup 2 --no-cfgA --no-cpu-baseline --no-parity --aux cfgC,cfgD,cfgE --aux-steps 8 --aux-warmup 3 > $OUT/bench_h$hv.out 2> $OUT/bench_h$hv.err
  cp bench_full.json $OUT/bench_h${hv}_full.json
  python - $OUT/bench_h${hv}_full.json $hv <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('halves', sys.argv[2], ' '.join('%s %.2f ms (flags %s, loss %.4f)' % (k, d[k]['ms_per_step'], d[k].get('cluster_handoff_flags'), d[k].get('final_loss', 0)) if isinstance(d.get(k), dict) and 'ms_per_step' in d[k] else '%s %s' % (k, d.get(k)) for k in ('cfgC', 'cfgD', 'cfgE')))
PY
done

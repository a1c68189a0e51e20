#!/bin/bash
# round 6: half-batch pipelines with more hardware queues (streams beyond GPU_MAX_HW_QUEUES share a queue and serialise)
set -u
OUT=${1:-gpurun_out/r06_halves2}
mkdir -p $OUT
export TMPDIR=/tmp
for q in 4 8; do for hv in 0 1; do
  GPU_MAX_HW_QUEUES=$q ASR_ENC_HALVES=$hv timeout 600 python bench.py --steps 5 --warmup 2 --no-cfgA --no-cpu-baseline --no-parity --aux cfgC,cfgD,cfgE --aux-steps 8 --aux-warmup 3 > $OUT/bench_q${q}_h$hv.out 2> $OUT/bench_q${q}_h$hv.err
  cp bench_full.json $OUT/bench_q${q}_h${hv}_full.json
  python - $OUT/bench_q${q}_h${hv}_full.json $q $hv <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('queues', sys.argv[2], 'halves', sys.argv[3], 'headline %.3f' % d['ms_per_step'], ' '.join('%s %.2f ms' % (k, d[k]['ms_per_step']) if isinstance(d.get(k), dict) and 'ms_per_step' in d[k] else '%s %s' % (k, d.get(k)) for k in ('cfgC', 'cfgD', 'cfgE')))
PY
done; done

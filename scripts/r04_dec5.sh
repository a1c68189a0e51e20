#!/bin/bash
# decoder cell products with bf16 weight images: tests, then cfg D / E timing with and without
set -u
OUT=gpurun_out/r04_dec5
mkdir -p $OUT
ASR_POISON_LDS=1 ASR_POISON_SCRATCH=1 timeout 600 python scripts/poison_pytest.py tests/test_gpu_attention.py -m gpu -q > $OUT/tests_att.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests_att.txt | cut -c1-250 | tail -12
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "cfgD or cfgE" > $OUT/tests_cfg.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests_cfg.txt | cut -c1-250 | tail -12
for v in 1 0; do
  ASR_DEC_CELL_BF16=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgD,cfgE > $OUT/b_$v.out 2> $OUT/b_$v.err
  tail -1 $OUT/b_$v.out | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cell_bf16=$v', {k: d[k]['ms_per_step'] for k in ('cfgD','cfgE') if k in d})" || tail -5 $OUT/b_$v.err
  python - <<PY
import json
d=json.load(open('bench_full.json'))
for k in ('cfgD','cfgE'):
    g=d.get(k,{}).get('greedy_infer')
    if g: print('  ', k, 'greedy_infer', {x: g[x] for x in g if x in ('tokens_per_s','ms','steps')})
PY
done

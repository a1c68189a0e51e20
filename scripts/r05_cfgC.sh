#!/bin/bash
# round 5: cfg C after the convolution changes: parity + determinism tests, bench (cfgC entry only), kernel trace
set -u
OUT=${1:-gpurun_out/r05_cfgC}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_determinism.py -q -x -k "conv or vgg or pool or cfgC" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cfgA --aux cfgC --no-cpu-baseline > $OUT/bench.out 2> $OUT/bench.err
cp bench_full.json $OUT/bench_full.json
python - <<PY
import json
d=json.load(open('$OUT/bench_full.json'))
e=d['cfgC']
print('cfgC', e['ms_per_step'], e['value'], e.get('mfma_frac_whole_step'), e['roofline']['kernel'], e['roofline']['frac'], (e.get('parity') or {}).get('loss_rel_err_vs_oracle'))
print({k:round(v['avg_us'],1) for k,v in e['kernels'].items()})
PY
bash scripts/r02_trace_cfgC.sh $OUT/cfgC > $OUT/cfgC.log 2>&1
head -24 $OUT/cfgC/stats.md | cut -c1-150

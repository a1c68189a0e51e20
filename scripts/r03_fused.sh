#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_fused}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_gpu_configs.py -q -x -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for V in 1 0; do
ASR_ATT_FUSED=$V timeout 400 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgD,cfgE --aux-steps 6 > $OUT/aux$V.json 2> $OUT/aux$V.err
python - <<PY
import json
d = json.load(open('$OUT/aux$V.json'))
print('fused=$V', ' '.join('%s %.2f ms' % (k, d[k]['ms_per_step']) for k in ('cfgD', 'cfgE') if d.get(k)))
PY
done

#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_colsum}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -q -x -m gpu -k "colsum or cfgC" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC --aux-steps 6 > $OUT/aux.json 2> $OUT/aux.err
python - <<PY
import json
d = json.load(open('$OUT/aux.json'))
print('cfgC %.2f ms' % d['cfgC']['ms_per_step'])
PY

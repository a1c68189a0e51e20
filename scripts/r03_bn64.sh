#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_bn64}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_model.py -q -x -m gpu -k "conv or cfgC or vgg" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
for V in 1 0; do
ASR_CONV_WGRAD_BN64=$V timeout 400 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC --aux-steps 6 > $OUT/aux$V.json 2> $OUT/aux$V.err
python - <<PY
import json
d = json.load(open('$OUT/aux$V.json'))
print('bn64=$V cfgC %.2f ms' % d['cfgC']['ms_per_step'])
PY
done

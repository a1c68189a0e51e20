#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_gemm2}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -q -x -m gpu -k "gemm or cfgC or cfgD_joint" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for V in 1 0; do
ASR_GEMM_NT_BIG=$V timeout 400 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC,cfgD,cfgE > $OUT/aux$V.json 2> $OUT/aux$V.err
python - <<PY
import json
d = json.load(open('$OUT/aux$V.json'))
print('big=$V headline %.3f ms' % d['ms_per_step'], ' '.join('%s %.2f ms' % (k, d[k]['ms_per_step']) for k in ('cfgC', 'cfgD', 'cfgE') if d.get(k)))
PY
done

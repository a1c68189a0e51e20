#!/bin/bash
# round 5: fp32 recurrence on the bf16 matrix pipe (three-term split): parity tests, then cfg A-shaped and 5x256 fp32 steps A/B
set -u
OUT=${1:-gpurun_out/r05_f32split}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "f32" > $OUT/tests_f32.txt 2>&1
tail -15 $OUT/tests_f32.txt | cut -c1-300
A="--units 128 --layers 2 --classes 39 --dtype f32 --keep-prob 0.5 --no-aux --no-cfgA --no-cpu-baseline --steps 30 --warmup 5"
for v in 1 0; do
  ASR_LSTM_F32_SPLIT=$v python bench.py $A > $OUT/cfgA_split$v.out 2> $OUT/cfgA_split$v.err
  python - <<PY
import json
d=json.loads(open('$OUT/cfgA_split$v.out').read().strip().splitlines()[-1])
print('cfgA split=$v', d['ms_per_step'], d['kernels'], d.get('parity'))
PY
done
A2="--units 256 --layers 5 --classes 61 --dtype f32 --no-aux --no-cfgA --no-cpu-baseline --no-parity --steps 10 --warmup 3"
for v in 1 0; do
  ASR_LSTM_F32_SPLIT=$v python bench.py $A2 > $OUT/f32_5x256_split$v.out 2> $OUT/f32_5x256_split$v.err
  python - <<PY
import json
d=json.loads(open('$OUT/f32_5x256_split$v.out').read().strip().splitlines()[-1])
print('5x256 f32 split=$v', d['ms_per_step'], d['kernels'])
PY
done

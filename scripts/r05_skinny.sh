#!/bin/bash
# round 5: decoder-step products with the activations split into hi + lo bf16 (ASR_SKINNY_SPLIT): tests + cfg D shaped step A/B
set -u
OUT=${1:-gpurun_out/r05_skinny}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_determinism.py -q -x -k "att or joint or cfgD or cfgE or bahdanau or seq2seq or decoder or cell or skinny" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt | cut -c1-300
for A in 0 1; do
  ASR_SKINNY_SPLIT=$A python scripts/probe_cfgD.py 2>&1 | grep "^it" | tail -2 | cut -c1-150 | sed "s/^/split=$A /" | tee -a $OUT/ab.txt
done
PREV=zeros bash scripts/r05_cfgD_tl.sh $OUT/tl > /dev/null 2>&1
grep "skinny" $OUT/tl/stats.md | cut -c1-200

#!/bin/bash
# round 5: attention backward's batched products on the bf16 pipe (ASR_ATT_BWD_BF16): parity tests + cfg D / E shaped step A/B
set -u
OUT=${1:-gpurun_out/r05_att}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_determinism.py -q -x -k "att or joint or cfgD or cfgE or bahdanau or seq2seq or decoder" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt | cut -c1-300
PREV=zeros bash scripts/r05_cfgD_tl.sh $OUT/tl > /dev/null 2>&1
for A in 0 1; do
  ASR_ATT_BWD_BF16=$A python scripts/probe_cfgD.py 2>&1 | grep "^it" | tail -2 | cut -c1-150 | sed "s/^/bwd_bf16=$A /" | tee -a $OUT/ab.txt
done

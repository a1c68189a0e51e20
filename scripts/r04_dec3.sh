#!/bin/bash
set -u
OUT=gpurun_out/r04_dec
mkdir -p $OUT
timeout 600 python scripts/poison_pytest.py tests/test_gpu_attention.py -m gpu -q > $OUT/tests_att_poison.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests_att_poison.txt | cut -c1-200 | tail -15
for i in 1 2 3 4 5 6; do
  timeout 120 python -m pytest tests/test_gpu_attention.py -m gpu -q -x -k "test_attention_model_parity and bahdanau" 2>&1 | tail -1 | cut -c1-120
done

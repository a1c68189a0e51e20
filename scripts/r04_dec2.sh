#!/bin/bash
set -u
OUT=gpurun_out/r04_dec
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q > $OUT/tests_att_a.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests_att_a.txt | cut -c1-200 | tail -15
timeout 300 python -m pytest tests/test_gpu_attention.py -m gpu -q -k "test_attention_model_parity" > $OUT/tests_att_b.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests_att_b.txt | cut -c1-200 | tail -8

#!/bin/bash
set -u
OUT=gpurun_out/r04_dec
mkdir -p $OUT
ASR_POISON_SCRATCH=1 timeout 900 python scripts/poison_pytest.py tests/test_gpu_attention.py tests/test_gpu_ops.py -m gpu -q > $OUT/tests_poison2.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests_poison2.txt | cut -c1-200 | tail -25

#!/bin/bash
# GPU tests with every kind of "memory nobody wrote" turned into NaN: torch allocator, work arena, LDS
set -u
OUT=gpurun_out/r04_poison
mkdir -p $OUT
ASR_POISON_LDS=1 ASR_POISON_SCRATCH=1 timeout 1200 python scripts/poison_pytest.py tests/test_gpu_attention.py tests/test_gpu_ops.py -m gpu -q > $OUT/tests.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests.txt | cut -c1-200 | tail -40

#!/bin/bash
# the whole -m gpu suite with the allocator, the work arena and every CU's LDS poisoned with NaN patterns
set -u
OUT=gpurun_out/r06_poison_all
mkdir -p $OUT
( time ASR_POISON_LDS=1 ASR_POISON_SCRATCH=1 timeout 1500 python scripts/poison_pytest.py tests -m gpu -q -k "not bench_last and not two_rank and not bare_gpus" > $OUT/tests.txt 2>&1 ) 2> $OUT/time.txt
grep -E "^FAILED|passed|failed" $OUT/tests.txt | cut -c1-250 | tail -15; grep real $OUT/time.txt

#!/bin/bash
set -u
OUT=gpurun_out/r04_order
mkdir -p $OUT
rm -f $OUT/peek.txt
ASR_PEEK_STICKY=$OUT/peek.txt timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "not rccl and not two_rank and not recipe and not bench" > $OUT/d.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/d.txt | cut -c1-200 | tail -6
cat $OUT/peek.txt 2>/dev/null | head -20

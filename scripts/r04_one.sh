#!/bin/bash
set -u
OUT=gpurun_out/r04_one
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "handoff_timeout or loss_grads_and_step" > $OUT/t1.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/t1.txt | cut -c1-200 | tail -3
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "not rccl and not two_rank and not recipe and not bench" > $OUT/all.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/all.txt | cut -c1-200 | tail -3

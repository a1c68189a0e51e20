#!/bin/bash
# does the injected hand-off timeout of test_gpu_ops.py leak into the cluster tests that follow it in another order?
set -u
OUT=gpurun_out/r04_order
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "handoff_timeout or loss_grads_and_step or test_ctc_model_train" > $OUT/a.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/a.txt | cut -c1-200 | tail -4
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -x -q -k "not rccl and not two_rank and not long" > $OUT/b.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/b.txt | cut -c1-200 | tail -4

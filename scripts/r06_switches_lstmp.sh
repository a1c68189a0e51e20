#!/bin/bash
# round 6 (late): the projected-cell (num_proj) paths and the gradient-blocking clip of asr_lstm_bwd_ex under the switches that
# select other kernels for them
set -u
OUT=gpurun_out/r06_switches_lstmp
mkdir -p $OUT
K="gradient_blocking_clip or lstmcell_projection or projected_layer or vgg_front_end_over_projected or multitask_ctc_parity or cldnn_ctc_model_parity"
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -k "$K" > $OUT/$name.txt 2>&1 )
  echo "$name: $(grep -E 'passed|failed' $OUT/$name.txt | tail -1)"
}
run default X=1
run fused0 ASR_LSTMP_FUSED=0
run bf16_0 ASR_LSTMP_BF16=0
run f32split0 ASR_LSTM_F32_SPLIT=0
run dflags512 ASR_LSTM_DFLAGS=512
run dflags16 ASR_LSTM_DFLAGS=16
run f32cluster0 ASR_LSTM_CLUSTER_F32=0
run cluster0 ASR_LSTM_CLUSTER=0

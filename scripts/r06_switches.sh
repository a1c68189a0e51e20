#!/bin/bash
# round 6: the recurrence / CTC / encoder tests under the A-B switches that select the other kernel paths
export TMPDIR=/tmp
OUT=gpurun_out/r06_switches; mkdir -p $OUT
run() { name=$1; shift; ( env "$@" timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "lstm or ctc or blstm or headline or cluster or model_step or encoder" > $OUT/$name.txt 2>&1 ); echo "$name: $(grep -E 'passed|failed|error' $OUT/$name.txt | tail -1)"; }
run default A=1
run write_through_exchange ASR_LSTM_DFLAGS=16
run f32_exact_mfma ASR_LSTM_F32_SPLIT=0
run ctc_one_wave ASR_CTC_WAVES=1
run enc_halves ASR_ENC_HALVES=1
run no_early ASR_LSTM_DFLAGS=32
run eight_wave_clusters ASR_LSTM_DFLAGS=512
( ASR_GRU_CLUSTER=0 timeout 900 python -m pytest tests -m gpu -q -x -k "gru" > $OUT/gru_single_cu.txt 2>&1 ); echo "gru_single_cu: $(grep -E 'passed|failed|error' $OUT/gru_single_cu.txt | tail -1)"
( timeout 900 python -m pytest tests -m gpu -q -x -k "gru" > $OUT/gru_clusters.txt 2>&1 ); echo "gru_clusters: $(grep -E 'passed|failed|error' $OUT/gru_clusters.txt | tail -1)"

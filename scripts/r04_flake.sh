#!/bin/bash
# how often does the first GPU test of the suite fail in a fresh process?
set -u
OUT=gpurun_out/r04_flake
mkdir -p $OUT
n=0; f=0
for i in $(seq 1 30); do
  timeout 120 python -m pytest tests/test_gpu_attention.py -m gpu -q -x -k "test_attention_model_parity and bahdanau_content" > $OUT/run_$i.txt 2>&1
  n=$((n+1))
  if ! grep -q " passed" $OUT/run_$i.txt || grep -q "failed" $OUT/run_$i.txt; then f=$((f+1)); echo "run $i FAILED"; grep -E "^E   " $OUT/run_$i.txt | head -3 | cut -c1-200; else rm -f $OUT/run_$i.txt; fi
done
echo "runs $n failures $f"

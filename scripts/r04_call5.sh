#!/bin/bash
set -u
OUT=gpurun_out/r04_call5
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "long" > $OUT/tests.txt 2>&1
grep -n "^cfg \|cfg D\|oracle vs\|passed\|failed\|Error\|whole gradient\|attention " $OUT/tests.txt | cut -c1-600
tail -3 $OUT/tests.txt | cut -c1-300

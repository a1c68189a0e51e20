#!/bin/bash
set -u
OUT=gpurun_out/r04_call3
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "lstm or infeasible or smoke or cfgB" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
bash scripts/r04_ab_flags.sh $OUT/ab "0 2048" 2>&1 | tee $OUT/ab.txt

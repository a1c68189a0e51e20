#!/bin/bash
set -u
OUT=gpurun_out/r04_call4
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "exchange_paths or headline_shapes" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
bash scripts/r04_ab_flags.sh $OUT/ab "0" 2>&1 | tee $OUT/ab.txt

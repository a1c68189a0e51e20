#!/bin/bash
# the full -m gpu suite in the driver's order with -x, on the final tree
set -u
OUT=gpurun_out/r04_suite
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
grep -E "^FAILED|passed|failed" $OUT/gpu_tests.txt | cut -c1-300 | tail -8; grep real $OUT/gpu_tests.time

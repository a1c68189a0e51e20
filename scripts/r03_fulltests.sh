#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_full}
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/gpu_tests.log 2>&1
tail -5 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

#!/bin/bash
# round 3, call 1: the configuration-width parity tests, the 2-rank bench line, the default bench line
set -u
OUT=${1:-gpurun_out/r03_c1}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py "tests/test_gpu_model.py::test_bench_line_of_a_two_rank_run" \
   "tests/test_gpu_model.py::test_bucketed_gradient_averaging_on_the_communication_stream" -q -s --durations=10 > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
echo "bench rc=$? $(grep real $OUT/bench.time)"
tail -3 $OUT/bench.err

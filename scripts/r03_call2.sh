#!/bin/bash
# round 3, call 2: new tests, bounded default bench (progress on stderr), recurrence phase timers at H = 256 / 512
set -u
OUT=${1:-gpurun_out/r03_c2}
mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_configs.py -k "cfgC" "tests/test_gpu_model.py::test_bucketed_gradient_averaging_on_the_communication_stream" \
   tests/test_gpu_attention.py -k "cfgC or bucketed or class_surface or bridge_classes" -q -s --durations=5 > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"
( time timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
echo "bench rc=$? $(grep real $OUT/bench.time)"
grep "bench " $OUT/bench.err | tail -40
for H in 256 512; do timeout 120 python scripts/probe_lstm_phases.py $H > $OUT/phases_$H.txt 2>&1; echo "phases $H rc=$?"; done

#!/bin/bash
# round 5, call 1: new tests first (determinism across fresh processes, padded saved activations, non-finite report,
# merge_repeated doc example), then the whole -m gpu suite, smoke, the default bench
set -u
OUT=${1:-gpurun_out/r05_call1}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_determinism.py -q -x > $OUT/determinism.txt 2>&1 ) 2> $OUT/determinism.time
tail -5 $OUT/determinism.txt | cut -c1-400; grep real $OUT/determinism.time
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "padded_frames or nonfinite or tensorflow_known or handoff_timeout" > $OUT/new_ops.txt 2>&1
tail -3 $OUT/new_ops.txt | cut -c1-400
( time timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_determinism.py > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
tail -3 $OUT/gpu_tests.txt | cut -c1-300; grep real $OUT/gpu_tests.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench default rc=$? $(grep real $OUT/bench_default.time)"; cp bench_full.json $OUT/bench_default_full.json
tail -1 $OUT/bench_default.out | wc -c
tail -1 $OUT/bench_default.out | cut -c1-1500

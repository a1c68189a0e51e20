#!/bin/bash
# round 5, last confirmation of HEAD: full -m gpu suite, smoke, default bench line, and the VGG / scheduling tests under poison
set -u
OUT=${1:-gpurun_out/r05_last}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
tail -2 $OUT/gpu_tests.txt | cut -c1-200; grep real $OUT/gpu_tests.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench default rc=$? $(grep real $OUT/bench_default.time)"; cp bench_full.json $OUT/bench_default_full.json
tail -1 $OUT/bench_default.out | cut -c1-400
ASR_POISON_LDS=1 ASR_POISON_SCRATCH=1 timeout 600 python scripts/poison_pytest.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -k "vgg or cfgC or scheduling or attention_backward" > $OUT/poison.txt 2>&1
tail -2 $OUT/poison.txt | cut -c1-200

#!/bin/bash
# round 6, first call: the new tests (fp32 cfg D / E at their own widths, bare --gpus 2, deferred-check depth), the whole
# -m gpu suite, smoke, the driver-settings bench line
set -u
OUT=${1:-gpurun_out/r06_call1}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "fp32_at_its_own or long_sequences" -s > $OUT/new_cfg_tests.txt 2>&1 ) 2> $OUT/new_cfg_tests.time
tail -3 $OUT/new_cfg_tests.txt | cut -c1-300; grep real $OUT/new_cfg_tests.time
( time timeout 1800 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
tail -3 $OUT/gpu_tests.txt | cut -c1-300; grep real $OUT/gpu_tests.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench20.out 2> $OUT/bench20.err ) 2> $OUT/bench20.time
echo "bench20 rc=$? $(grep real $OUT/bench20.time)"; cp bench_full.json $OUT/bench20_full.json
tail -1 $OUT/bench20.out | wc -c

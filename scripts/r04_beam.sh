#!/bin/bash
set -u
OUT=gpurun_out/r04_beam
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "beam or decode or greedy" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt | cut -c1-300
timeout 600 python bench.py --steps 3 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux decode > $OUT/bench.out 2> $OUT/bench.err
tail -1 $OUT/bench.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['decode'], indent=0))"

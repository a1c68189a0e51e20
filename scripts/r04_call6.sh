#!/bin/bash
set -u
OUT=gpurun_out/r04_call6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/tests.txt 2>&1
tail -15 $OUT/tests.txt | cut -c1-300
timeout 600 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgD,cfgE > $OUT/bench.out 2> $OUT/bench.err
tail -1 $OUT/bench.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('cfgD','cfgE')})"
python -c "
import json; d=json.load(open('bench_full.json')); print(d['cfgD'].get('greedy_infer')); print(d['cfgE'].get('greedy_infer'))"

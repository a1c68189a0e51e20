#!/bin/bash
set -u
OUT=gpurun_out/r04_cfgC
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "cfgC or vgg or full_chip" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt | cut -c1-200
timeout 300 python bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC > $OUT/b.out 2> $OUT/b.err
tail -1 $OUT/b.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cfgC'])"

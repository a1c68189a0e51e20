#!/bin/bash
# cfg C (VGG + 4x512, B = 64): workgroups of the weight-gradient GEMMs that run beside a BPTT kernel (ASR_BG_WGS)
set -u
OUT=gpurun_out/r04_bgwgs
mkdir -p $OUT
for W in 128 256 512 1024; do
  ASR_BG_WGS=$W timeout 300 python bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC,cfgD > $OUT/w$W.out 2> $OUT/w$W.err
  tail -1 $OUT/w$W.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BG_WGS $W', {k:(d[k]['ms_per_step'], d[k]['kernel_us']['lstm_bwd']) for k in ('cfgC','cfgD')})"
done

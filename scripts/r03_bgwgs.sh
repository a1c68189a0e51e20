#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_bgwgs3}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
for V in 64 96 128 192 256; do
  ASR_BG_WGS=$V timeout 120 python bench.py --steps 20 --warmup 5 $Q > $OUT/b256_w$V.json 2>> $OUT/err.log
done
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    d = json.load(open(p)); k = d['kernels']
    print('%-18s %.3f ms/step fwd %.1f bwd %.1f' % (p.split('/')[-1], d['ms_per_step'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us']))
PY

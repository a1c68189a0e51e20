#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_bgc}
mkdir -p $OUT
for V in 128 256 512 1024; do
ASR_BG_WGS=$V timeout 400 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC,cfgD --aux-steps 5 > $OUT/aux$V.json 2> $OUT/aux$V.err
python - <<PY
import json
d = json.load(open('$OUT/aux$V.json'))
print('bg_wgs=$V cfgC %.2f ms cfgD %.2f ms' % (d['cfgC']['ms_per_step'], d['cfgD']['ms_per_step']))
PY
done

#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_probe2}
mkdir -p $OUT
timeout 300 python scripts/probe_matmul.py 2>&1 | grep -v Warning | tee $OUT/matmul.txt
timeout 400 python bench.py --steps 5 --warmup 2 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC,cfgD,cfgE > $OUT/aux.json 2> $OUT/aux.err
python - <<PY
import json
d = json.load(open('$OUT/aux.json'))
for k in ('cfgC', 'cfgD', 'cfgE'):
    e = d.get(k)
    if e: print(k, {x: e[x] for x in e if x in ('value', 'ms_per_step', 'unit')}, (e.get('kernels') or {}).get('lstm_fwd'), (e.get('kernels') or {}).get('lstm_bwd'))
PY
tail -3 $OUT/aux.err

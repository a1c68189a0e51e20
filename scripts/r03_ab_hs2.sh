#!/bin/bash
# A/B of the 4-wave cluster members: forward / backward / both, exclusive-CU LDS padding on / off
set -u
OUT=${1:-gpurun_out/r03_hs2}
mkdir -p $OUT
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --steps 20 --warmup 5 $Q > $OUT/b256_$name.json 2> $OUT/b256_$name.err
  env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --units 512 --batch 32 $Q > $OUT/b512_$name.json 2> $OUT/b512_$name.err
}
run base X=1
run fwd32 ASR_LSTM_FWD_HS=32

run bwd32 ASR_LSTM_BWD_HS=32
run both32 ASR_LSTM_HS=32
ASR_LSTM_HS=32 timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "cluster" > $OUT/tests_hs32.log 2>&1
echo "cluster tests with HS=32: $(tail -1 $OUT/tests_hs32.log)"
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/b*.json')):
    try:
        d = json.load(open(p)); k = d['kernels']
        print('%-32s %.0f frames/s %.3f ms/step (median %.3f host %.2f) fwd %.1f bwd %.1f us handoff %s loss %.4f' % (p.split('/')[-1], d['value'], d['ms_per_step'],
              d['step_ms']['median'], d['step_ms']['host_issue_mean'], k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags'], d['final_loss']))
    except Exception as e:
        print(p, 'ERR', e)
PY

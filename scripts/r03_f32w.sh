#!/bin/bash
# fp32 cluster kernels at H = 128 / 256 / 320 / 512: parity tests, which kernels ran, and step-time A/B against the
# single-CU fp32 kernels (ASR_LSTM_CLUSTER_F32_WIDE=0).   usage: r03_f32w.sh [OUT]
set -u
OUT=${1:-gpurun_out/r03_f32w}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "cluster_f32 or lstm_fwd_f32 or lstm_bwd_f32 or bf16_and_wide" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_ops.py -q -x -m gpu -k "cluster_f32 and (320 or 512)" > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
for p in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'lstm' in r['Name']:
            print(r['Name'][:110], r['Calls'], r['AverageNs'])
PY
Q="--no-aux --no-cfgA --no-parity --no-cpu-baseline --dtype f32"
run() {  # name env... -- args
  local name=$1; shift
  env "$@" timeout 200 python bench.py $Q $ARGS > $OUT/$name.json 2> $OUT/$name.err
}
ARGS="--steps 10 --warmup 3 --units 128 --layers 2 --classes 39 --keep-prob 0.5"; run a128 X=1
ARGS="--steps 5 --warmup 2 --units 256 --layers 5"; run w256_on X=1; run w256_off ASR_LSTM_CLUSTER_F32_WIDE=0
ARGS="--steps 5 --warmup 2 --units 320 --layers 5"; run w320_on X=1; run w320_off ASR_LSTM_CLUSTER_F32_WIDE=0
ARGS="--steps 4 --warmup 2 --units 512 --layers 5 --batch 32"; run w512_on X=1; run w512_e ASR_LSTM_DFLAGS=32; run w512_off ASR_LSTM_CLUSTER_F32_WIDE=0
python - <<PY
import json, glob
for p in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.load(open(p)); k = d['kernels']
        print('%-16s %.0f frames/s %.3f ms/step fwd %.1f bwd %.1f us handoff %s loss %.5f' % (p.split('/')[-1], d['value'], d['ms_per_step'],
              k['lstm_fwd']['avg_us'], k['lstm_bwd']['avg_us'], d['cluster_handoff_flags'], d['final_loss']))
    except Exception as e:
        print(p, 'ERR', e)
PY

#!/bin/bash
# data-parallel bench path with 2 ranks on a 1-GPU box: both ranks on cuda:0, gloo instead of RCCL (RCCL refuses two
# ranks on one device).  Exercises the launcher contract, the per-layer bucketed averaging on the communication stream,
# the barrier / max-over-ranks timing and the rank-0 JSON line -- not the xGMI collective itself.
# usage: bash scripts/r02_dp_dryrun.sh <outdir>
set -u
OUT=${1:-gpurun_out/r02_dp}
mkdir -p $OUT
export ASR_BENCH_DEVICE=0 ASR_BENCH_BACKEND=gloo ASR_DP_COLLECTIVE=torch
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_dp2.json 2> $OUT/bench_dp2.err
echo "rc=$?"
tail -3 $OUT/bench_dp2.err
python - <<PY
import json
d = json.load(open('$OUT/bench_dp2.json'))
print('n_gpus', d['n_gpus'], 'value %.0f frames/s' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'parallelism', d['config']['parallelism'],
      'global_batch', d['config']['global_batch'], 'final_loss', d['final_loss'], 'handoff', d['cluster_handoff_flags'])
PY

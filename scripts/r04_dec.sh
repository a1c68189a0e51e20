#!/bin/bash
# decoder step: product + cell in one launch, last-arriver combine -- tests, then cfg D / E timing with each switch
set -u
OUT=gpurun_out/r04_dec
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -x -q > $OUT/tests_att.txt 2>&1
tail -3 $OUT/tests_att.txt | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "cfgD or cfgE" > $OUT/tests_cfg.txt 2>&1
tail -3 $OUT/tests_cfg.txt | cut -c1-250
for v in "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  ASR_DEC_CELL_GEMM=$1 ASR_ATT_TAIL=$2 timeout 300 python bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgD,cfgE > $OUT/b_$1$2.out 2> $OUT/b_$1$2.err
  tail -1 $OUT/b_$1$2.out | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cellgemm=$1 tail=$2', {k: (d[k]['ms_per_step'], d[k].get('greedy_infer')) for k in ('cfgD','cfgE') if k in d})" || tail -5 $OUT/b_$1$2.err
done

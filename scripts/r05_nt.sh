#!/bin/bash
# round 5: non-temporal loads on the attention loops' streams (build-time ASR_ATT_NT): tests + cfg D shaped step + kernel stats
set -u
OUT=${1:-gpurun_out/r05_nt}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention.py -q -x > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt | cut -c1-300
python scripts/probe_cfgD.py 2>&1 | grep "^it" | tail -2 | cut -c1-150 | tee -a $OUT/ab.txt
PREV=zeros bash scripts/r05_cfgD_tl.sh $OUT/tl > /dev/null 2>&1
grep "skinny\|att_" $OUT/tl/stats.md | cut -c1-60,120-200

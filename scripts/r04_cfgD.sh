#!/bin/bash
# round 4: parity of the decoder-loop kernels + kernel trace of the cfg-D-shaped joint CTC-attention step
set -u
OUT=${1:-gpurun_out/r04_cfgD}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_configs.py -m gpu -x -q -k "not long" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt | cut -c1-300
rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgD -- python scripts/probe_cfgD.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
head -2 $OUT/timeline.md
tail -3 $OUT/probe.log
head -22 $OUT/stats.md | cut -c1-180
rm -rf $OUT/trace

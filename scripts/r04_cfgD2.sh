#!/bin/bash
# kernel trace of the cfg-D-shaped step after the decoder-step changes (product + cell in one launch, bf16 weight images)
set -u
OUT=gpurun_out/r04_cfgD2
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgD -- python scripts/probe_cfgD.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
tail -3 $OUT/probe.log
head -24 $OUT/stats.md | cut -c1-200
rm -rf $OUT/trace

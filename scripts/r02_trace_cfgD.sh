#!/bin/bash
# kernel trace of the cfg-D-shaped joint CTC-attention step (scripts/probe_cfgD.py)
set -u
OUT=${1:-gpurun_out/r02_cfgD}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgD -- python scripts/probe_cfgD.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
head -2 $OUT/timeline.md
tail -3 $OUT/probe.log
head -40 $OUT/stats.md | cut -c1-160
rm -rf $OUT/trace

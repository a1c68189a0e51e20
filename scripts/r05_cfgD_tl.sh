#!/bin/bash
# round 5: cfg-D-shaped joint CTC-attention step: kernel trace + timeline of one step
set -u
OUT=${1:-gpurun_out/r05_cfgD_tl}
mkdir -p $OUT
export TMPDIR=/tmp
PREV=${PREV:-zeros} rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgD -- python scripts/probe_cfgD.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md ${MARKER:-optimizer_kernel} > /dev/null
grep "^it" $OUT/probe.log | tail -3
rm -rf $OUT/trace

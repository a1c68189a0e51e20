#!/bin/bash
# kernel trace of the headline bench (rocprofv3 --kernel-trace --stats) + one-step timeline
set -u
OUT=${1:-gpurun_out/r02_trace}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-cfgA > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
echo "db: $DB"
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
tail -3 $OUT/bench_trace.log | cut -c1-400
head -5 $OUT/timeline.md
rm -rf $OUT/trace

#!/bin/bash
# round 3: kernel trace of the headline bench (rocprofv3 --kernel-trace --stats) + one-step timeline
set -u
OUT=${1:-gpurun_out/r03_trace}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
echo "db: $DB"
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
tail -2 $OUT/bench_trace.log | cut -c1-300
head -5 $OUT/timeline.md
rm -rf $OUT/trace

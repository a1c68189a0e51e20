#!/bin/bash
# round 5: cfg-C-shaped step: kernel trace + TIMELINE of one step (what runs beside what; is the TN weight-gradient GEMM exposed?)
set -u
OUT=${1:-gpurun_out/r05_tn}
mkdir -p $OUT
export TMPDIR=/tmp
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
grep cfgC $OUT/probe.log | tail -3
rm -rf $OUT/trace
python scripts/bench_tn.py > $OUT/bench_gemm.txt 2>&1
tail -30 $OUT/bench_gemm.txt

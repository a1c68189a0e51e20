#!/bin/bash
# round 5: weight-gradient lanes behind the dx product (ASR_DW_AFTER_DX) x high-priority main stream, cfg C shape; timeline of the best
set -u
OUT=${1:-gpurun_out/r05_sched}
mkdir -p $OUT
export TMPDIR=/tmp
for A in 0 ""; do for P in "" -1; do
  ASR_DW_AFTER_DX=$A MAIN_PRIO=$P ONLY_C=1 python scripts/probe_cfgCE.py 2>&1 | grep cfgC | tail -2 | sed "s/^/after_dx '$A' prio '$P' /" | tee -a $OUT/ab.txt
done; done
MAIN_PRIO=${TRACE_PRIO:-} ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
rm -rf $OUT/trace

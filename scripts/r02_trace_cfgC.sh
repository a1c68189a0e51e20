#!/bin/bash
# kernel trace of the cfg-C-shaped VGG-BLSTM CTC step (scripts/probe_cfgCE.py, beam part skipped)
set -u
OUT=${1:-gpurun_out/r02_cfgC}
mkdir -p $OUT
export TMPDIR=/tmp
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
grep cfgC $OUT/probe.log | tail -3
head -34 $OUT/stats.md | cut -c1-150
rm -rf $OUT/trace

#!/bin/bash
# round 3: matrix-core utilisation and LDS bank conflicts per kernel (rocprofv3 --pmc, ONE counter per pass, kernel trace
# only next to --pmc) on the headline step and on the cfg-C-shaped step.   usage: r03_pmc_util.sh [OUT]
set -u
OUT=${1:-gpurun_out/r03_pmc_util}
mkdir -p $OUT
export TMPDIR=/tmp
for c in MfmaUtil SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/h_$c -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/h_$c.log 2>&1
  DB=$(find $OUT/h_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py "$DB" > $OUT/headline_$c.txt
  rm -rf $OUT/h_$c
  ONLY_C=1 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/c_$c -o pmc -- python scripts/probe_cfgCE.py > $OUT/c_$c.log 2>&1
  DB=$(find $OUT/c_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py "$DB" > $OUT/cfgC_$c.txt
  rm -rf $OUT/c_$c
  echo "== $c"; head -6 $OUT/headline_$c.txt | cut -c1-140; head -8 $OUT/cfgC_$c.txt | cut -c1-140
done

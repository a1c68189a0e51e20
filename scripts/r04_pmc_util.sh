#!/bin/bash
# round 4, final tree: matrix-core utilisation per kernel of the cfg-C-shaped step (rocprofv3 --pmc MfmaUtil, kernel trace only)
set -u
OUT=gpurun_out/r04_pmc_util
mkdir -p $OUT
export TMPDIR=/tmp
for c in MfmaUtil; do
  ONLY_C=1 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/c_$c -o pmc -- python scripts/probe_cfgCE.py > $OUT/c_$c.log 2>&1
  DB=$(find $OUT/c_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py "$DB" > $OUT/cfgC_$c.txt
  rm -rf $OUT/c_$c
  grep -E "conv|gemm_nt|gemm_tn|pool" $OUT/cfgC_$c.txt | cut -c1-150
done

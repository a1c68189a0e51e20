#!/bin/bash
# round 6, final tree: per-kernel counters of the cfg-C-shaped step -- matrix-core utilisation, LDS bank conflicts / LDS
# instruction time / busy cycles, and the HBM bytes of the image-resident weight-gradient kernel (FETCH_SIZE / WRITE_SIZE).
# One rocprofv3 --pmc pass per counter group, kernel trace only (no other trace domain beside --pmc).
set -u
OUT=${1:-gpurun_out/r06_pmc_cfgC}
mkdir -p $OUT
export TMPDIR=/tmp
for c in MfmaUtil SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS FETCH_SIZE WRITE_SIZE; do
  ONLY_C=1 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/c_$c -o pmc -- python scripts/probe_cfgCE.py > $OUT/c_$c.log 2>&1
  DB=$(find $OUT/c_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py "$DB" > $OUT/cfgC_$c.txt
  rm -rf $OUT/c_$c
  echo "== $c"; grep -E "conv3x3|gemm_nt|gemm_tn|pool|lstm" $OUT/cfgC_$c.txt | cut -c1-160 | head -16
done

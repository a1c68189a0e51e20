#!/bin/bash
# round 4, final tree: LDS bank conflicts / LDS instruction time / busy cycles per kernel of the cfg-C-shaped step
set -u
OUT=gpurun_out/r04_pmc_util2
mkdir -p $OUT
export TMPDIR=/tmp
for c in SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS; do
  ONLY_C=1 timeout 120 rocprofv3 --pmc $c --kernel-trace -d $OUT/c_$c -o pmc -- python scripts/probe_cfgCE.py > $OUT/c_$c.log 2>&1
  DB=$(find $OUT/c_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py "$DB" > $OUT/cfgC_$c.txt
  rm -rf $OUT/c_$c
  echo "== $c"; grep -E "conv3x3_img|wgrad_tr" $OUT/cfgC_$c.txt | cut -c1-150
done

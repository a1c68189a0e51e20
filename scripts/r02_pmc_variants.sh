#!/bin/bash
# HBM-side traffic of the recurrence kernels of the cfg-A-shaped fp32 step and the 5x320 step (same recipe as r02_pmc.sh)
set -u
OUT=${1:-gpurun_out/r02_pmc_variants}
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 40 rocprofv3 --pmc $c --kernel-trace -d $OUT/${name}_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-cfgA "$@" > $OUT/${name}_$c.log 2>&1
    DB=$(find $OUT/${name}_$c -name '*.db' | head -1)
    python scripts/rocpd_pmc.py "$DB" lstm_ > $OUT/${name}_$c.txt
    cut -c1-150 $OUT/${name}_$c.txt
    rm -rf $OUT/${name}_$c
  done
}
run cfgA --units 128 --layers 2 --dtype f32 --classes 39 --keep-prob 0.5
run h320 --units 320

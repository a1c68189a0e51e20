#!/bin/bash
# round 4: HBM-side traffic of the headline step's kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE
# passes (one counter per pass, kernel trace only next to --pmc), per-kernel averages by scripts/rocpd_pmc.py.
# usage: r04_pmc.sh OUT [extra bench args, e.g. "--units 512 --batch 32"]
set -u
OUT=${1:-gpurun_out/r04_pmc}
EXTRA=${2:-}
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-cfgA --no-aux $EXTRA > $OUT/$c.log 2>&1
  DB=$(find $OUT/$c -name '*.db' | head -1)
  python scripts/rocpd_pmc.py "$DB" > $OUT/$c.txt
  head -8 $OUT/$c.txt | cut -c1-150
  rm -rf $OUT/$c
done

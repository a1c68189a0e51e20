#!/bin/bash
# kernel traces (rocprofv3 --kernel-trace --stats) of the cfg-A-shaped fp32 step and the 5x320 bf16 step
set -u
OUT=${1:-gpurun_out/r02_trace_variants}
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/$name -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-cfgA "$@" > $OUT/$name.log 2>&1
  DB=$(find $OUT/$name -name '*.db' | head -1)
  python scripts/rocpd_stats.py "$DB" $OUT/${name}_stats.md > /dev/null
  head -12 $OUT/${name}_stats.md | cut -c1-160
  rm -rf $OUT/$name
}
run cfgA --units 128 --layers 2 --dtype f32 --classes 39 --keep-prob 0.5
run h320 --units 320

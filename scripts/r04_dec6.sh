#!/bin/bash
set -u
OUT=gpurun_out/r04_dec6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_ops.py -m gpu -q -x > $OUT/tests.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests.txt | cut -c1-250 | tail -5
timeout 300 python bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgD,cfgE > $OUT/b.out 2> $OUT/b.err
tail -1 $OUT/b.out | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k: d[k]['ms_per_step'] for k in ('cfgD','cfgE') if k in d})" || tail -5 $OUT/b.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgD -- python scripts/probe_cfgD.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
grep -E "skinny|cell|att_" $OUT/stats.md | cut -c1-60,110-190
rm -rf $OUT/trace

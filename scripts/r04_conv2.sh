#!/bin/bash
set -u
OUT=gpurun_out/r04_conv2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "not rccl and not two_rank" > $OUT/tests.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/tests.txt | cut -c1-220 | tail -6
timeout 300 python bench.py --steps 10 --warmup 3 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC > $OUT/b.out 2> $OUT/b.err
tail -1 $OUT/b.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cfgC'])" || tail -5 $OUT/b.err
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
grep -E "conv|pool|gemm_nt|gemm_tn|colsum|dropout" $OUT/stats.md | cut -c1-70,110-170
rm -rf $OUT/trace

#!/bin/bash
# round 6: reducers with eight slab loads in flight -- GEMM / convolution parity tests (bit-identical sums), determinism,
# cfg C / D / E of the bench, kernel trace of the cfg C step
set -u
OUT=${1:-gpurun_out/r06_reduce}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_determinism.py -m gpu -q -x -k "gemm or conv or vgg or determin or wgrad" > $OUT/tests.txt 2>&1 ) 2> $OUT/tests.time
tail -2 $OUT/tests.txt | cut -c1-300; grep real $OUT/tests.time
timeout 600 python bench.py --steps 10 --warmup 3 --no-cfgA --no-cpu-baseline --no-parity --aux cfgC,cfgD,cfgE --aux-steps 8 --aux-warmup 3 > $OUT/bench.out 2> $OUT/bench.err
cp bench_full.json $OUT/bench_full.json
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('headline %.3f' % d['ms_per_step'], ' '.join('%s %.2f ms' % (k, d[k]['ms_per_step']) if isinstance(d.get(k), dict) and 'ms_per_step' in d[k] else '%s %s' % (k, d.get(k)) for k in ('cfgC', 'cfgD', 'cfgE')))
PY
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/traceC -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probeC.log 2>&1
DB=$(find $OUT/traceC -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/cfgC_stats.md > /dev/null; python scripts/rocpd_timeline.py "$DB" $OUT/cfgC_timeline.md > /dev/null
rm -rf $OUT/traceC
grep -i "reduce" $OUT/cfgC_stats.md | cut -c1-160

#!/bin/bash
# round 4: the driver's round-end sequence -- full -m gpu suite, smoke, default bench (compact last line)
set -u
OUT=gpurun_out/r04_full
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
tail -3 $OUT/gpu_tests.txt | cut -c1-300; grep real $OUT/gpu_tests.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench rc=$? $(grep real $OUT/bench_default.time)"
tail -c 8000 $OUT/bench_default.out | tail -1 | python -c "import sys,json; l=sys.stdin.read(); print(len(l)); d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value']); print({k:(d[k]['ms_per_step'],d[k]['host_issue_mean'],d[k]['host_wait_for_device_mean']) for k in ('cfgA','cfgC','cfgD','cfgE') if k in d and 'ms_per_step' in d[k]})"
cp bench_full.json $OUT/bench_full_default.json

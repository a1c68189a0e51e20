#!/bin/bash
# round 4, call 1: compact bench line as the driver runs it + the new model-level gpu tests
set -u
OUT=gpurun_out/r04_call1
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench20.out 2> $OUT/bench20.err ) 2> $OUT/bench20.time
echo "bench20 rc=$? $(grep real $OUT/bench20.time)"
tail -c 8000 $OUT/bench20.out | tail -1 | python -c "import sys,json; l=sys.stdin.read(); print(len(l)); d=json.loads(l); print({k:(d[k]['ms_per_step'],d[k]['host_issue_mean'],d[k]['host_wait_for_device_mean']) for k in ('cfgA','cfgC','cfgD','cfgE') if k in d}); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
cp bench_full.json $OUT/bench_full20.json
timeout 1500 python -m pytest tests -m gpu -x -q -k "bench or infeasible or cfg or joint or attention or vgg or gru" > $OUT/tests_subset.txt 2>&1
tail -5 $OUT/tests_subset.txt

#!/bin/bash
# round 4, final tree (c): the driver's round-end sequence (full -m gpu suite, smoke, default bench), the 20 / 5 bench line,
# kernel traces of the cfg-C-shaped step and of the headline step
set -u
OUT=gpurun_out/r04_full3
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
grep -E "^FAILED|passed|failed" $OUT/gpu_tests.txt | cut -c1-300 | tail -8; grep real $OUT/gpu_tests.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench rc=$? $(grep real $OUT/bench_default.time)"
tail -c 8000 $OUT/bench_default.out | tail -1 | python -c "import sys,json; l=sys.stdin.read(); print(len(l)); d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value']); print({k:(d[k]['ms_per_step'],d[k]['host_issue_mean'],d[k]['host_wait_for_device_mean']) for k in ('cfgA','cfgC','cfgD','cfgE') if k in d and 'ms_per_step' in d[k]})"
cp bench_full.json $OUT/bench_full_default.json
( time timeout 900 python bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench20.out 2> $OUT/bench20.err ) 2> $OUT/bench20.time
echo "bench20 rc=$? $(grep real $OUT/bench20.time)"; cp bench_full.json $OUT/bench20_full.json
tail -1 $OUT/bench20.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels'))"
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/cfgC_stats.md > /dev/null
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
head -4 $OUT/timeline.md; head -6 $OUT/stats.md | cut -c1-160
rm -rf $OUT/trace

#!/bin/bash
# round 6 evidence run: the driver's sequence (full -m gpu suite incl. the determinism tests, smoke, default bench) +
# driver-settings bench line + kernel trace / timeline of the headline + PMC HBM traffic (headline, cfg A shape) +
# kernel traces of cfg A / cfg C / cfg D
set -u
OUT=${1:-gpurun_out/r06_final}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1 ) 2> $OUT/gpu_tests.time
tail -3 $OUT/gpu_tests.txt | cut -c1-300; grep real $OUT/gpu_tests.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench20.out 2> $OUT/bench20.err ) 2> $OUT/bench20.time
echo "bench20 rc=$? $(grep real $OUT/bench20.time)"; cp bench_full.json $OUT/bench20_full.json
( time timeout 900 python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench default rc=$? $(grep real $OUT/bench_default.time)"; cp bench_full.json $OUT/bench_default_full.json
tail -1 $OUT/bench_default.out | wc -c
# kernel trace + timeline of the headline
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-cfgA --no-aux > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
head -5 $OUT/timeline.md; head -8 $OUT/stats.md | cut -c1-160
rm -rf $OUT/trace
# cfg A shape
A="--units 128 --layers 2 --classes 39 --dtype f32 --keep-prob 0.5"
rocprofv3 --kernel-trace --stats -d $OUT/traceA -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-cfgA --no-aux $A > $OUT/benchA_trace.log 2>&1
DB=$(find $OUT/traceA -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/statsA.md > /dev/null
head -8 $OUT/statsA.md | cut -c1-160
rm -rf $OUT/traceA
bash scripts/r04_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
bash scripts/r04_pmc.sh $OUT/pmcA "$A" > $OUT/pmcA.log 2>&1
grep "lstm_fwd\|lstm_bwd\|optimizer" $OUT/pmc/*.txt $OUT/pmcA/*.txt | cut -c1-220
# cfg C / cfg D shaped steps: kernel statistics + timeline of one step
mkdir -p $OUT/cfgC $OUT/cfgD
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/cfgC/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/cfgC/probe.log 2>&1
DB=$(find $OUT/cfgC/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/cfgC/stats.md > /dev/null; python scripts/rocpd_timeline.py "$DB" $OUT/cfgC/timeline.md > /dev/null
rm -rf $OUT/cfgC/trace
rocprofv3 --kernel-trace --stats -d $OUT/cfgD/trace -o cfgD -- python scripts/probe_cfgD.py > $OUT/cfgD/probe.log 2>&1
DB=$(find $OUT/cfgD/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/cfgD/stats.md > /dev/null; python scripts/rocpd_timeline.py "$DB" $OUT/cfgD/timeline.md > /dev/null
rm -rf $OUT/cfgD/trace
grep cfgC $OUT/cfgC/probe.log | tail -1; grep "^it" $OUT/cfgD/probe.log | tail -1 | cut -c1-120
ls $OUT

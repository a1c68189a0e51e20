#!/bin/bash
# round 3 evidence run: driver-settings bench line (all entries), kernel trace + timeline of the headline, PMC HBM
# traffic (headline, 5x512 B=32, the cfg-A-shaped fp32 step, 5x320), kernel traces of the cfg C / cfg D shaped steps
set -u
OUT=${1:-gpurun_out/r03_final}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 420 python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err ) 2> $OUT/bench20.time
echo "bench20 rc=$? $(grep real $OUT/bench20.time)"
grep "bench " $OUT/bench20.err | tail -30
( time timeout 420 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench default rc=$? $(grep real $OUT/bench_default.time)"
bash scripts/r03_trace.sh $OUT/trace > $OUT/trace.log 2>&1
bash scripts/r03_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
bash scripts/r03_pmc.sh $OUT/pmc512 "--units 512 --batch 32" > $OUT/pmc512.log 2>&1
bash scripts/r03_pmc.sh $OUT/pmcA "--units 128 --layers 2 --dtype f32 --classes 39 --keep-prob 0.5" > $OUT/pmcA.log 2>&1
bash scripts/r03_pmc.sh $OUT/pmc320 "--units 320" > $OUT/pmc320.log 2>&1
bash scripts/r02_trace_cfgC.sh $OUT/cfgC > $OUT/cfgC.log 2>&1
bash scripts/r02_trace_cfgD.sh $OUT/cfgD > $OUT/cfgD.log 2>&1
ls $OUT $OUT/trace $OUT/pmc

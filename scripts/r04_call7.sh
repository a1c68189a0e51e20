#!/bin/bash
set -u
OUT=gpurun_out/r04_call7
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "lstm or handoff or smoke or cfgB or beam" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt | cut -c1-200
bash scripts/r04_ab_flags.sh $OUT/ab "0" 2>&1 | tee $OUT/ab.txt
bash scripts/r04_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
grep "lstm_fwd\|lstm_bwd\|optimizer" $OUT/pmc/*.txt | cut -c1-200

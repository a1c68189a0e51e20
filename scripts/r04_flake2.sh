#!/bin/bash
set -u
OUT=gpurun_out/r04_flake2
mkdir -p $OUT
n=0; f=0
for i in $(seq 1 45); do
  timeout 60 python scripts/flake_probe.py $i > $OUT/p_$i.txt 2>&1 || { f=$((f+1)); cat $OUT/p_$i.txt | tail -12; }
  n=$((n+1))
done
echo "runs $n failures $f"; head -1 $OUT/p_1.txt

#!/bin/bash
# round 5: workgroups of the weight-gradient GEMMs beside a BPTT kernel (ASR_BG_WGS), now that they all start behind the dx product
set -u
OUT=${1:-gpurun_out/r05_bgwgs}
mkdir -p $OUT
for W in 64 128 256 512; do
  ASR_BG_WGS=$W ONLY_C=1 python scripts/probe_cfgCE.py 2>&1 | grep cfgC | tail -2 | sed "s/^/bg_wgs $W /" | tee -a $OUT/ab.txt
done
for W in 128 512; do
  ASR_BG_WGS=$W python scripts/probe_cfgD.py 2>&1 | grep "^it" | tail -1 | cut -c1-120 | sed "s/^/cfgD bg_wgs $W /" | tee -a $OUT/ab.txt
done

#!/bin/bash
# round 5: image-resident weight gradient: parity tests, timing (new vs ASR_CONV_WGRAD_IMG=0)
set -u
OUT=${1:-gpurun_out/r05_wgrad}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv3x3_implicit or vgg" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt | cut -c1-300
ASR_CONV_DBG=0 python scripts/probe_conv_phases.py 2>&1 | grep "N=\|weight gradient" | tee $OUT/new.txt
ASR_CONV_WGRAD_IMG=0 ASR_CONV_DBG=0 python scripts/probe_conv_phases.py 2>&1 | grep "N=\|weight gradient" | tee $OUT/old.txt

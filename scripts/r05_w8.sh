#!/bin/bash
set -u
OUT=${1:-gpurun_out/r05_w8}
mkdir -p $OUT
export TMPDIR=/tmp
ASR_CONV_IMG_W8=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "conv3x3_implicit or conv_relu_dropout" > $OUT/tests.txt 2>&1
tail -2 $OUT/tests.txt | cut -c1-200
ASR_CONV_IMG_W8=1 ASR_CONV_DBG=0 python scripts/probe_conv_phases.py 2>&1 | grep "N=\|ACT" | tee $OUT/w8.txt
ASR_CONV_DBG=0 python scripts/probe_conv_phases.py 2>&1 | grep "N=\|ACT" | tee $OUT/w4.txt

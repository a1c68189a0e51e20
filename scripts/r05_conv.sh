#!/bin/bash
# round 5: image-resident convolution with the deferred / interleaved epilogue: parity tests, production timing, phase timers
set -u
OUT=${1:-gpurun_out/r05_conv}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -k "conv or vgg or pool" > $OUT/tests_conv.txt 2>&1
tail -5 $OUT/tests_conv.txt | cut -c1-300
ASR_CONV_DBG=0 python scripts/probe_conv_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/prod.txt
ASR_CONV_DBG=1 python scripts/probe_conv_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dbg.txt

#!/bin/bash
# round 5: convolution changes in the model: VGG parity tests at the cfg C widths, the cfg-C-shaped step (trace)
set -u
OUT=${1:-gpurun_out/r05_conv2}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_determinism.py -q -x -k "conv or vgg or pool or cfgC" > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt | cut -c1-300
ASR_CONV_DBG=0 python scripts/probe_conv_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/prod.txt
bash scripts/r02_trace_cfgC.sh $OUT/cfgC > $OUT/cfgC.log 2>&1
grep cfgC $OUT/cfgC/probe.log | tail -3
head -16 $OUT/cfgC/stats.md | cut -c1-150

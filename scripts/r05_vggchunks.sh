#!/bin/bash
# round 5: VGG forward in runs of images on separate lanes (ASR_VGG_FWD_CHUNKS): tests + cfg C A/B + timeline
set -u
OUT=${1:-gpurun_out/r05_vggchunks}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_determinism.py -q -x -k "vgg or cfgC or scheduling" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt | cut -c1-300
for A in 1 2 3 1 2; do
  ASR_VGG_FWD_CHUNKS=$A ONLY_C=1 python scripts/probe_cfgCE.py 2>&1 | grep cfgC | tail -1 | sed "s/^/chunks=$A /" | tee -a $OUT/ab.txt
done
ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
rm -rf $OUT/trace

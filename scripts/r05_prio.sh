#!/bin/bash
# round 5: does a high-priority main stream keep the side lanes' weight-gradient GEMMs out of the dx GEMM's way? (cfg C shape)
set -u
OUT=${1:-gpurun_out/r05_prio}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.Stream.priority_range())" > $OUT/range.txt 2>&1
for P in "" -1; do
  MAIN_PRIO=$P ONLY_C=1 python scripts/probe_cfgCE.py 2>&1 | grep cfgC | tail -2 | sed "s/^/prio '$P' /" | tee -a $OUT/ab.txt
done
MAIN_PRIO=-1 ONLY_C=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o cfgC -- python scripts/probe_cfgCE.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" $OUT/timeline.md > /dev/null
rm -rf $OUT/trace
cat $OUT/range.txt

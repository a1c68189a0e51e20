#!/bin/bash
set -u
OUT=${1:-gpurun_out/r05_ntp}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or tiles or nt" > $OUT/tests.txt 2>&1
tail -2 $OUT/tests.txt | cut -c1-200
for P in 0 256 512; do
  ASR_GEMM_NT_PERSIST=$P python scripts/bench_nt.py 2>&1 | grep "^NT" | sed "s/^/persist $P /" | tee -a $OUT/ab.txt
done
for P in 0 256; do
  ASR_GEMM_NT_PERSIST=$P ONLY_C=1 python scripts/probe_cfgCE.py 2>&1 | grep cfgC | tail -2 | sed "s/^/persist $P /" | tee -a $OUT/ab.txt
done

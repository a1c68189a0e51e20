#!/bin/bash
set -u
OUT=${1:-gpurun_out/r03_gemm}
mkdir -p $OUT
for V in 1 0; do
  echo "== ASR_GEMM_NT_BIG=$V"
  SKIP_LIB=1 ASR_GEMM_NT_BIG=$V timeout 200 python scripts/probe_matmul.py 2>&1 | grep -E "xproj" | tee -a $OUT/gemm_v$V.txt
done

#!/bin/bash
# fused ReLU/pool + dropout epilogues: bit-identity tests, the VGG model tests, cfg C timing with and without
set -u
OUT=gpurun_out/r04_drop
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -s -k "epilogue or pool or vgg or cfgC or dropout or conv3x3" > $OUT/tests.txt 2>&1
grep -E "fused vs separate|passed|failed|Error" $OUT/tests.txt | cut -c1-220 | tail -12
for f in 1 0; do
  ASR_VGG_FUSED_DROP=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC > $OUT/b$f.out 2> $OUT/b$f.err
  tail -1 $OUT/b$f.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$f', d['cfgC'])" || tail -5 $OUT/b$f.err
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o cfgC -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cfgA --no-parity --no-cpu-baseline --aux cfgC > /root/repo/$OUT/prof.out 2>&1
cd /root/repo
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200 > $OUT/kernel_stats_head.txt

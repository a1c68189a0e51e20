#!/bin/bash
# kernel trace of the cfg-E-shaped CTC beam decode (T = 1000, C = 3387, width 100)
set -u
OUT=${1:-gpurun_out/r02_beam}
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/beam_probe.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tensorflow_end2end_speech_recognition_amd import ops
dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
for (T, C, W, B) in [(1000, 3387, 100, 1), (1000, 3387, 100, 8), (1000, 3387, 100, 32)]:
    logits = torch.tensor(rng.randn(T, B, C) * 3, dtype=torch.float32, device=dev)
    sl = torch.full((B,), T, dtype=torch.int32, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lab, n, sc = ops.ctc_beam_decode(logits, sl, beam_width=W)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print('beam T=%d C=%d W=%d B=%d: %.1f ms (%.1f us/frame)' % (T, C, W, B, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / T), flush=True)
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace -o beam -- python /tmp/beam_probe.py > $OUT/probe.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $OUT/stats.md > /dev/null
grep "^beam" $OUT/probe.log
head -12 $OUT/stats.md | cut -c1-170
rm -rf $OUT/trace

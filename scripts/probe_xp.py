#!/usr/bin/env python
"""Round 4 debugging aid: one bf16 LSTM layer forward + BPTT through the cluster kernels for (H, B, T, flags) given on
the command line, one subprocess per case (a GPU fault aborts the process); prints max |difference| of dgates against
the flag-2048 (unpaired reduce-scatter slots) run of the same case and the hand-off error word."""
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import test_gpu_ops as t
from tensorflow_end2end_speech_recognition_amd import ops
H, B, T, flags = [int(v) for v in sys.argv[1:5]]
rng = np.random.RandomState(7)
lens = rng.randint(1, T + 1, size=B); lens[0] = T
x, ps = t._lstm_case(rng, T, B, 24, H, 2, lens, init=0.1)
dout = rng.randn(T, B, 2 * H)
cuda = torch.device('cuda', 0)
ops.debug_set_lstm_flags(2048)
ref = t._run_hip_layer(cuda, x, ps, lens, H, 2, 'bf16', 50.0, dout)
e0 = ops.check_async_errors(0)
ops.debug_set_lstm_flags(flags)
got = t._run_hip_layer(cuda, x, ps, lens, H, 2, 'bf16', 50.0, dout)
e1 = ops.check_async_errors(0)
print('H=%%d B=%%d T=%%d flags=%%d: dgates diff %%.3e (ref max %%.3e) nan %%d  err %%d/%%d' %% (H, B, T, flags, np.abs(got['dgates'] - ref['dgates']).max(),
      np.abs(ref['dgates']).max(), int(np.isnan(got['dgates']).sum()), e0, e1))
''' % (ROOT, ROOT)
for case in sys.argv[1:]:
    r = subprocess.run([sys.executable, '-c', CHILD] + case.split(','), capture_output=True, text=True, timeout=300)
    print(case, 'rc', r.returncode, r.stdout.strip()[-300:])
    if r.returncode != 0:
        print('   stderr head:', r.stderr[:600].replace('\n', ' | '))

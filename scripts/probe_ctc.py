#!/usr/bin/env python
"""asr_ctc_loss at the label lengths of the BASELINE configurations: time per call (HIP events) with the alpha / beta
recursions on one wave (ASR_CTC_WAVES=1) and on 2 - 4 waves (default).  Shapes: headline (T 778, L 75 -> 3 states per lane,
always one wave), cfg C (T 1650, L 235, B 64), cfg D (T 1600, L 400, B 32), cfg E (T 1000, L 166, C 3387, B 32)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from tensorflow_end2end_speech_recognition_amd import ops

dev = torch.device('cuda:0')
for name, T, B, C, L in (('headline', 778, 16, 62, 75), ('cfgC', 1650, 64, 29, 235), ('cfgD', 1600, 32, 29, 400),
                         ('cfgE', 1000, 32, 3387, 166)):
    rng = np.random.RandomState(T)
    logits = torch.tensor(rng.randn(T, B, C).astype(np.float32), device=dev)
    sl = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    sl[0] = T
    labs = [[int(v) for v in rng.randint(0, C - 1, size=min(L, n // 3))] for n in sl]
    labs[0] = [int(v) for v in rng.randint(0, C - 1, size=L)]
    flat = np.asarray(sum(labs, []), dtype=np.int32)
    off = np.zeros(B + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(l) for l in labs])
    args = (logits, torch.tensor(flat, device=dev), torch.tensor(off, device=dev), torch.tensor(sl, device=dev), L)
    row = []
    for waves in ('1', None):
        if waves:
            os.environ['ASR_CTC_WAVES'] = waves
        else:
            os.environ.pop('ASR_CTC_WAVES', None)
        for _ in range(3):
            ops.ctc_loss(*args, grad_scale=1.0)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record()
            ops.ctc_loss(*args, grad_scale=1.0)
            b.record()
        torch.cuda.synchronize()
        row.append(sorted(a.elapsed_time(b) for a, b in ev)[5] * 1e3)
    print('%-9s T %4d B %2d C %4d L %3d: asr_ctc_loss %.0f us on one wave, %.0f us on several' % (name, T, B, C, L, row[0], row[1]))

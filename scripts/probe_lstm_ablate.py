#!/usr/bin/env python
"""Ablation table of the headline recurrence kernels (lstm_{fwd,bwd}_cluster8_kernel<256, ..., 32>): launch time with one
piece of the per-step chain left out at a time (template parameter ABL, library built with ASR_BUILD_ABLATE=1; the ablated
kernels compute garbage -- only their duration is looked at).  What a piece costs ON the serial chain is the difference to
the full kernel; the sum of the pieces against the full step shows how much of the step overlaps.

    ASR_BUILD_ABLATE=1 python -m tensorflow_end2end_speech_recognition_amd.build --force     # here
    gpurun -- python scripts/probe_lstm_ablate.py > gpurun_out/ablate.txt                        # on the box
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from bench import make_batch
from tensorflow_end2end_speech_recognition_amd import ops
from tensorflow_end2end_speech_recognition_amd._lib import ASR_BF16

H, B, D, ndir = int(os.environ.get('ABL_H', '256')), int(os.environ.get('ABL_B', '16')), 120, 2
dev = torch.device('cuda:0')
x, sl, labels, dense = make_batch(1, B, D, 62, 100, int(os.environ.get('ABL_T', '778')))     # the bench's batch: T = 778, 49 % padding
T = x.shape[1]
g = torch.Generator(device='cpu').manual_seed(0)
xd = ops.bt_to_tb(torch.tensor(x, device=dev), ASR_BF16)
xproj = torch.empty((T, B, ndir * 4 * H), dtype=torch.float32, device=dev)
whf = torch.empty((ndir, 4 * H * H), dtype=torch.bfloat16, device=dev)
whb = torch.empty_like(whf)
for d in range(ndir):
    kernel = (torch.rand((D + H, 4 * H), generator=g) * 0.2 - 0.1).to(dev)
    bias = torch.zeros(4 * H, device=dev)
    w = ops.lstm_prep_weights(kernel, bias, D, H, ASR_BF16)
    ops.gemm(xd.view(T * B, -1)[:, :D].contiguous() if xd.shape[2] != D else xd.view(T * B, D), w['wx_il'], bias=w['bias_il'],
             out=xproj.view(T * B, -1)[:, d * 4 * H:(d + 1) * 4 * H])
    whf[d].copy_(w['pf'])
    whb[d].copy_(w['pb'])
peep = (torch.rand((ndir, 3, H), generator=g) * 0.2 - 0.1).to(dev)
sld = torch.tensor(sl, dtype=torch.int32, device=dev)
dout = torch.randn((T, B, ndir * H), generator=g).to(dev)


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2] * 1e3, ms[0] * 1e3


gates, hout, cs, cf, hf = ops.lstm_fwd(xproj, whf, peep, sld, H, ndir, ASR_BF16, 1.0, 50.0)
torch.cuda.synchronize()
FW = ['poll: no wait for valid tags', 'no A-fragment LDS reads', 'no MFMAs', 'no gate math', 'no saved-activation stores',
      'no barriers', 'no x-projection fetch', 'no publish', 'no LDS staging of polled slices', 'no poll loads at all']
BW = ['poll: no wait for valid tags', 'no A-fragment LDS reads', 'no MFMAs', 'no gate-gradient math', 'no dgates store',
      'no barrier', 'no saved-activation fetch', 'no publish stores', 'no dG write to LDS', 'no poll loads at all']


def name(bits, names):
    return ' + '.join(names[k] for k in range(10) if (bits >> k) & 1) or 'full kernel'


print('T = %d, B = %d, H = %d, ndir = %d; us per launch (median of 12, min), us per recurrence step' % (T, B, H, ndir))
for label, env, names, variants, fn in (
        ('forward', 'ASR_LSTM_ABL_FWD', FW, (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 6, 14, 48, 513, 515, 519, 527, 545, 769, 800, 1023) if H == 256
         else (0, 1, 8, 16, 64, 512, 6, 513, 519, 527, 545, 1023),
         lambda: ops.lstm_fwd(xproj, whf, peep, sld, H, ndir, ASR_BF16, 1.0, 50.0)),
        ('BPTT', 'ASR_LSTM_ABL_BWD', BW, (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 6, 14, 80, 144, 513, 515, 519, 527, 545, 641, 1023) if H == 256
         else (0, 1, 8, 64, 512, 6, 513, 519, 527, 545, 641, 1023),
         lambda: ops.lstm_bwd(dout, gates, cs, whb, peep, sld, H, ndir, ASR_BF16))):
    base = None
    print('\n| %s ABL | left out | us / launch (median) | min | us / step | delta vs full (us / step) |' % label)
    print('|---|---|---|---|---|---|')
    for v in variants:
        if v:
            os.environ[env] = str(v)
        else:
            os.environ.pop(env, None)
        med, mn = timed(fn)
        if base is None:
            base = med
        print('| %d | %s | %.1f | %.1f | %.3f | %+.3f |' % (v, name(v, names), med, mn, med / T, (med - base) / T))
    os.environ.pop(env, None)
    # hand-off error word: ablated kernels may time out by construction; clear it
    try:
        ops.check_async_errors(0)
    except Exception as e:
        print('(error word after the %s ablations: %s)' % (label, str(e)[:100]))

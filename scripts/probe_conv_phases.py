"""Phase timers of the image-resident 3x3 convolution (ASR_CONV_DBG=1: conv3x3_img_kernel<..., DBG=true>).
usage: ASR_CONV_DBG=1 python scripts/probe_conv_phases.py   -> per shape: us per launch, cycles per image and wave by phase."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.getcwd())
os.environ.setdefault('ASR_CONV_DBG', '1')
import numpy as np
import torch
from tensorflow_end2end_speech_recognition_amd import _lib, ops

dev = torch.device('cuda:0')
lib = _lib.handle(0).lib
lib.asr_debug_conv_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = int(os.environ.get('NIMG', 59000))
g = torch.Generator(device=dev)
g.manual_seed(0)
for (H, W, Ci, Co) in ((40, 11, 64, 64), (20, 6, 64, 128), (20, 6, 128, 128)):
    x = torch.randn((N, H, W, Ci), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    w = torch.randn((3, 3, Ci, Co), generator=g, device=dev, dtype=torch.float32) * 0.05
    b = torch.zeros(Co, device=dev)
    wf, wb = ops.conv3x3_prep_weights(w)
    for _ in range(2):
        y = ops.conv3x3_fwd(x, wf, b, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = ops.conv3x3_fwd(x, wf, b, relu=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    buf = (ctypes.c_ulonglong * (64 * 4 * 8))()
    rc = lib.asr_debug_conv_cycles(buf, 64 * 4 * 8)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 8).astype(np.float64)
    flops = 2.0 * N * H * W * 9 * Ci * Co
    print('%dx%d %d->%d  N=%d: %.1f us  %.0f TFLOP/s  (timers rc %d)' % (H, W, Ci, Co, N, us, flops / us / 1e6, rc))
    if rc == 0:
        imgs = a[:, :, 5]
        per = a / np.maximum(imgs[:, :, None], 1)
        m = per.mean(axis=(0, 1))
        mx = per.max(axis=(0, 1))
        names = ['tile loop', '  multiplies + LDS reads', '  epilogues', 'next image: wait + LDS store', 'closing barrier',
                 'images per workgroup', 'prefetch issue']
        for k in (6, 0, 1, 2, 3, 4, 5):
            print('   %-32s mean %9.0f   max wave %9.0f' % (names[k], imgs.mean() if k == 5 else m[k], imgs.max() if k == 5 else mx[k]))
        print('   sum of phases per image (mean wave): %.0f cycles' % (m[6] + m[0] + m[3] + m[4]))
    # the other epilogues of the cfg C step: forward ReLU + dropout (ACT 3), data gradient gated by the dropped activation (ACT 2)
    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(); fn(); a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) * 1e3
    t3 = timed(lambda: ops.conv3x3_fwd_drop(x, wf, b, (0.8, 123, 0)))
    dy = torch.randn((N, H, W, Co), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    below = torch.relu(x)
    t2 = timed(lambda: ops.conv3x3_bwd_data_relu(dy, wb, below, drop=(0.8, 123, 0), dropped=True))
    fl2 = 2.0 * N * H * W * 9 * Ci * Co
    print('   fwd + dropout (ACT 3): %.1f us  %.0f TFLOP/s;  data gradient + gate (ACT 2, %d->%d): %.1f us  %.0f TFLOP/s'
          % (t3, flops / t3 / 1e6, Co, Ci, t2, fl2 / t2 / 1e6))
    dw = torch.zeros(9 * Ci, Co, device=dev)
    tw = timed(lambda: ops.conv3x3_bwd_weight(x, dy, dw))
    print('   weight gradient: %.1f us  %.0f TFLOP/s' % (tw, flops / tw / 1e6))
    del x, y, dy, below

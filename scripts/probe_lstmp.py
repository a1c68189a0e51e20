"""lstm_impl='LSTMCell' with num_proj (the projected cells): the fused layer (whole-sequence recurrence kernels on
W_p W_h, everything else batched) against the step-by-step layer (ASR_LSTMP_FUSED=0) -- one CTC training step on the
headline batch (B = 16, T <= 778, D = 120, 61 classes), fp32, loss and gradients side by side."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC  # noqa: E402

dev = torch.device('cuda:0')
for enc, H, P, L, B in (('blstm', 256, 128, 5, 16), ('blstm', 320, 160, 3, 16), ('lstm', 512, 256, 3, 32)):
    x, sl, _, dense = bench.make_batch(1, B, 120, 62, 100, 778)
    xd, sd = torch.tensor(x, device=dev), torch.tensor(sl, device=dev)
    res = {}
    for mode, steps in (('1', 10), ('0', 2), ('bf16', 10)):
        os.environ['ASR_LSTMP_FUSED'] = '0' if mode == '0' else '1'
        m = CTC(enc, 120, H, L, 61, lstm_impl='LSTMCell', num_proj=P, clip_grad_norm=5.0, clip_activation=50.0, seed=0,
                dtype='bf16' if mode == 'bf16' else 'f32', device=str(dev))
        loss, _ = m.compute_loss(xd, dense, sd, keep_prob=1.0)
        gv = m._set_optimizer('sgd', 0.1).compute_gradients(loss, model=m)
        grads = {v: g.detach().clone() for g, v in gv}
        l0 = float(loss.item())
        for _ in range(2):
            l_, _ = m.compute_loss(xd, dense, sd, keep_prob=0.8)
            m.train(l_, 'rmsprop', 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            l_, _ = m.compute_loss(xd, dense, sd, keep_prob=0.8)
            m.train(l_, 'rmsprop', 1e-3)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / steps, l0, grads)
        del m
    worst = max(float((res['1'][2][k] - res['0'][2][k]).abs().max() / max(float(res['0'][2][k].abs().max()), 1e-30))
                for k in res['0'][2])
    worst16 = max(float((res['bf16'][2][k] - res['0'][2][k]).abs().max() / max(float(res['0'][2][k].abs().max()), 1e-30))
                  for k in res['0'][2])
    print('%s %dx%d proj %d B=%d T=%d: fused %.2f ms/step (%.0f frames/s)  step-by-step %.1f ms/step  x%.1f   '
          'loss %.6f / %.6f   worst gradient difference (relative to the largest entry) %.1e   |   bf16 operands: %.2f ms/step '
          '(%.0f frames/s)  loss %.6f  worst gradient difference %.1e'
          % (enc, L, H, P, B, x.shape[1], res['1'][0] * 1e3, float(sl.sum()) / res['1'][0], res['0'][0] * 1e3,
             res['0'][0] / res['1'][0], res['1'][1], res['0'][1], worst, res['bf16'][0] * 1e3, float(sl.sum()) / res['bf16'][0],
             res['bf16'][1], worst16), flush=True)
os.environ.pop('ASR_LSTMP_FUSED')

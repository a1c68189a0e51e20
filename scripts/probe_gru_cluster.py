"""GRU forward: the cluster kernel against the single-CU persistent kernel (ASR_GRU_CLUSTER=0) -- values and time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tensorflow_end2end_speech_recognition_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
for H, B, T, ndir in ((128, 16, 60, 2), (256, 16, 60, 2), (256, 32, 45, 1), (128, 48, 33, 2), (256, 16, 381, 2), (320, 16, 381, 2), (64, 16, 381, 2)):
    xg = torch.tensor(rng.randn(T, B, ndir * 2 * H) * 0.5, dtype=torch.float32, device=dev)
    xc = torch.tensor(rng.randn(T, B, ndir * H) * 0.5, dtype=torch.float32, device=dev)
    wgh = torch.tensor(rng.randn(ndir, H, 2 * H) * 0.08, dtype=torch.float32, device=dev)
    wch = torch.tensor(rng.randn(ndir, H, H) * 0.08, dtype=torch.float32, device=dev)
    sl_np = rng.randint(1, T + 1, size=B).astype(np.int32)
    sl_np[0] = T
    if B > 16:
        sl_np[17] = 0
    sl = torch.tensor(sl_np, device=dev)
    res = {}
    for mode in ('0', '1'):
        os.environ['ASR_GRU_CLUSTER'] = mode
        out = ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 3, {k: v.clone() for k, v in out.items()})
    # backward on the persistent kernel's saved activations (the same inputs for both)
    saved = res['0'][1]
    dout = torch.tensor(rng.randn(T, B, ndir * H) * 0.3, dtype=torch.float32, device=dev)
    dhf = torch.tensor(rng.randn(ndir, B, H) * 0.3, dtype=torch.float32, device=dev)
    wghT = wgh.transpose(1, 2).contiguous()
    wchT = wch.transpose(1, 2).contiguous()
    bres = {}
    for mode in ('0', '1'):
        os.environ['ASR_GRU_CLUSTER'] = mode
        dg, dc = ops.gru_bwd(dout, dhf, saved, wghT, wchT, sl, T, H, ndir)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dg, dc = ops.gru_bwd(dout, dhf, saved, wghT, wchT, sl, T, H, ndir)
        torch.cuda.synchronize()
        bres[mode] = ((time.perf_counter() - t0) / 3, dg.clone(), dc.clone())
    flags = ops.check_async_errors(0)
    print('   backward: persistent %.3f ms  cluster %.3f ms (%.2f us/step)  max |diff| dgate %.1e (max %.1e)  dcand %.1e (max %.1e)'
          % (bres['0'][0] * 1e3, bres['1'][0] * 1e3, bres['1'][0] * 1e6 / T, float((bres['0'][1] - bres['1'][1]).abs().max()),
             float(bres['0'][1].abs().max()), float((bres['0'][2] - bres['1'][2]).abs().max()), float(bres['0'][2].abs().max())))
    worst = {k: float((res['0'][1][k] - res['1'][1][k]).abs().max()) for k in res['0'][1]}
    print('H=%d B=%d T=%d ndir=%d: persistent %.3f ms  cluster %.3f ms  (%.2f us/step)  max |diff| %s  error word %s'
          % (H, B, T, ndir, res['0'][0] * 1e3, res['1'][0] * 1e3, res['1'][0] * 1e6 / T,
             {k: '%.1e' % v for k, v in worst.items()}, flags), flush=True)
os.environ.pop('ASR_GRU_CLUSTER')

import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import make_batch
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
from tensorflow_end2end_speech_recognition_amd import ops
dev = torch.device('cuda:0')
for (tmax, H, L, dt) in [(778, 128, 1, 'bf16'), (778, 256, 1, 'bf16'), (778, 256, 5, 'bf16'), (778, 256, 5, 'f32')]:
    x, sl, labels, dense = make_batch(1, 16, 120, 62, min(100, tmax), tmax)
    m = CTC('blstm', 120, H, L, 61, clip_grad_norm=5.0, clip_activation=50, dtype=dt, seed=0)
    xd = torch.tensor(x, device=dev); sld = torch.tensor(sl, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _ = m.compute_loss(xd, dense, sld, keep_prob=0.8)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m.train(loss, 'rmsprop', 1e-3)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(tmax, H, L, dt, 'it', it, 'fwd %.1f ms  bwd+upd %.1f ms  loss %.3f' % ((t1-t0)*1e3, (t2-t1)*1e3, loss.item()), flush=True)
print('cpu count', os.cpu_count())

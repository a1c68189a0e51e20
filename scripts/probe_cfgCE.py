"""cfg C (VGG-BLSTM 4x512 CTC, B=64) training step and cfg E (C=3386, beam width 100) CTC beam decode."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tensorflow_end2end_speech_recognition_amd import ops
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
dev = torch.device('cuda:0')
rng = np.random.RandomState(2)
if os.environ.get('DO_C', '1') == '1':
    B, F, splice = int(os.environ.get('PB', 64)), 40, 11
    D = F * splice * 3
    tmax = int(os.environ.get('PT', 1650))
    sl = rng.randint(150, tmax + 1, size=B).astype(np.int32)
    T = int(sl.max())
    x = rng.randn(B, T, D).astype(np.float32)
    labs = [[int(v) for v in rng.randint(0, 28, size=max(1, n // 7))] for n in sl]
    dense = np.full((B, max(len(l) for l in labs)), -1, dtype=np.int64)
    for b, l in enumerate(labs):
        x[b, sl[b]:] = 0
        dense[b, :len(l)] = l
    m = CTC('vgg_blstm', F, 512, 4, 28, splice=splice, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=0) if False else None
    try:
        m = CTC('vgg_blstm', F * 3, 512, 4, 28, splice=splice, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=0)
        xd = torch.tensor(x, device=dev)
        if os.environ.get('MAIN_PRIO'):            # the step on a high-priority stream: side lanes yield the CUs to it
            torch.cuda.synchronize()
            torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ['MAIN_PRIO'])))
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss, _ = m.compute_loss(xd, dense, sl, keep_prob=0.8)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            m.train(loss, 'rmsprop', 1e-3)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print('cfgC it', it, 'B', B, 'T', T, 'fwd %.1f ms bwd %.1f ms loss %.2f -> %.0f frames/s' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, loss.item(), sl.sum() / (t2 - t0)), flush=True)
    except Exception as e:
        print('cfgC failed:', repr(e)[:300])
# cfg E: beam decode
for (T, C, W, B) in [] if os.environ.get('ONLY_C') else [(1000, 3387, 100, 1), (600, 62, 20, 16), (1000, 3387, 100, 8)]:
    logits = torch.tensor(rng.randn(T, B, C).astype(np.float32) * 3, device=dev)
    sl = torch.full((B,), T, dtype=torch.int32, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lab, n, sc = ops.ctc_beam_decode(logits, sl, beam_width=W)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print('beam T=%d C=%d W=%d B=%d: %.1f ms (%.1f us/frame)' % (T, C, W, B, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / T), flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lab, n = ops.ctc_greedy_decode(logits, sl)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('greedy T=%d C=%d B=%d: %.2f ms' % (T, C, B, (t1 - t0) * 1e3), flush=True)

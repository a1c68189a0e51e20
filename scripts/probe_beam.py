"""Prefix beam search: time per call / per frame at the bench's decode shapes for 256 and 512 threads per utterance
(ASR_BEAM_THREADS), results compared between the two.  ASR_BEAM_DBG=1 prints the phase cycles of utterance 0."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tensorflow_end2end_speech_recognition_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
rng = np.random.RandomState(5)
for name, T, B, C, W, shape in (('timit61_beam20', 778, 16, 62, 20, 'flat'), ('timit61_beam20_peaked', 778, 16, 62, 20, 'peaked'),
                                ('kanji3387_beam100', 1000, 8, 3387, 100, 'flat'),
                                ('kanji3387_beam100_peaked', 1000, 8, 3387, 100, 'peaked'),
                                ('kanji3387_beam20_peaked', 1000, 8, 3387, 20, 'peaked'),
                                ('c1000_beam40', 400, 8, 1000, 40, 'flat')):
    if shape == 'flat':
        lg = rng.randn(T, B, C).astype(np.float32) * 3
    else:
        lg = rng.randn(T, B, C).astype(np.float32)
        win = np.where(rng.rand(T, B) < 0.6, C - 1, rng.randint(0, C - 1, size=(T, B)))
        np.put_along_axis(lg, win[:, :, None], 12.0 + rng.rand(T, B, 1).astype(np.float32), axis=2)
    logits = torch.tensor(lg, device=dev)
    sl = torch.full((B,), T, dtype=torch.int32, device=dev)
    res = {}
    for nt in (256, 512):
        os.environ['ASR_BEAM_THREADS'] = str(nt)
        r = ops.ctc_beam_decode(logits, sl, beam_width=W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = ops.ctc_beam_decode(logits, sl, beam_width=W)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 3
        res[nt] = (t, [x.cpu().numpy() for x in r])
    same = all(np.array_equal(a, b) for a, b in zip(res[256][1][:2], res[512][1][:2]))
    ds = float(np.abs(res[256][1][2] - res[512][1][2]).max())
    print('%-26s 256: %7.2f ms (%5.1f us/frame)   512: %7.2f ms (%5.1f us/frame)   labels identical %s  |dscore| %.2e'
          % (name, res[256][0] * 1e3, res[256][0] / T * 1e6, res[512][0] * 1e3, res[512][0] / T * 1e6, same, ds), flush=True)
os.environ.pop('ASR_BEAM_THREADS')

"""Randomised differential check of asr_ctc_beam_decode against oracle.decoders.beam_search_decode: many small cases with
random vocabulary / width / length on smooth and peaked posteriors (exit status 1 on any difference).
FUZZ_TIES=1 adds quantised logits, repeated frames and constant posteriors -- inputs full of mathematically EXACT ties at the
beam cut.  There the device's fp64 log-softmax / logsumexp (its own summation order and libm) and numpy's can round a tied
pair of totals apart in one implementation and not in the other, which changes which of the tied candidates survive the
cut: round 6 saw 2 of 300 such cases differ (seed 1: a W = 100 cut at C = 12 on logits in steps of 0.5, and a repeated-frame
case), identically with the round-5 kernel, the device's answer having the higher full CTC likelihood both times.  Those are
reported, not counted as failures.  Usage: python scripts/fuzz_beam.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import ctc as octc  # noqa: E402
from oracle import decoders as odec  # noqa: E402
from tensorflow_end2end_speech_recognition_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
TIES = os.environ.get('FUZZ_TIES') == '1'
bad = 0
for case in range(n_cases):
    C = int(rng.choice([2, 3, 5, 12, 30, 62, 130, 400, 700]))
    W = int(rng.choice([1, 2, 5, 16, 20, 33, 64, 100, 128]))
    T = int(rng.randint(1, 14 if C * W > 20000 else 22))
    B = int(rng.randint(1, 4))
    kind = rng.choice(['smooth', 'ties', 'peaked', 'repeat', 'flat'] if TIES else ['smooth', 'peaked'])
    sharp = float(rng.choice([0.05, 0.5, 1.0, 3.0]))
    lg = (rng.randn(T, B, C) * sharp).astype(np.float32)
    if kind == 'ties':
        lg = np.round(lg * 2) / 2
    elif kind == 'peaked':
        win = np.where(rng.rand(T, B) < 0.6, C - 1, rng.randint(0, max(1, C - 1), size=(T, B)))
        np.put_along_axis(lg, win[:, :, None], 10.0 + rng.rand(T, B, 1).astype(np.float32), axis=2)
    elif kind == 'repeat':
        for t in range(1, T):
            if rng.rand() < 0.5:
                lg[t] = lg[t - 1]
    elif kind == 'flat':
        lg[:] = 0.0
    sl = rng.randint(0, T + 1, size=B).astype(np.int32)
    sl[0] = T
    lab, n, score = ops.ctc_beam_decode(torch.tensor(lg, device=dev), torch.tensor(sl, device=dev), W)
    lp = octc.log_softmax(lg.astype(np.float64).transpose(1, 0, 2))
    ref, rs = odec.beam_search_decode(lp, sl, C - 1, W)
    for b in range(B):
        got = lab[b, :int(n[b])].cpu().tolist()
        if got != ref[b] or (sl[b] > 0 and abs(score[b].item() - rs[b]) > 1e-7 * max(1, abs(rs[b]))):
            bad += 1
            print('MISMATCH case %d b %d: C=%d W=%d T=%d kind=%s sharp=%g\n  got %s\n  ref %s' % (case, b, C, W, T, kind, sharp, got, ref[b]))
print('%d cases, %d mismatches' % (n_cases, bad))
sys.exit(1 if (bad and not TIES) else 0)

"""One fresh-process run of the first GPU test of the suite (test_attention_model_parity[bahdanau_content]) with its
intermediate results compared one by one, for the unreproduced failure recorded in DESIGN section 2 (f): prints one line,
and on a mismatch dumps which stage is off (encoder output, keys, attention weights, logits per step).
Usage: python scripts/flake_probe.py <tag>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
torch.set_num_threads(4)
from oracle import attention as oatt  # noqa: E402
import test_gpu_attention as tga  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else '?'
att = 'bahdanau_content'
rng = np.random.RandomState(11)
B, T, D, H, L, U, A, Em, C = 5, 17, 12, 64, 1, 128, 32, 8, 9
x, sl, labels, lsl, _ = tga._batch(rng, B, T, D, C)
model = tga._mk(AttentionSeq2Seq, att, D, H, L, U, A, Em, C, sharpening_factor=1.5, logits_temperature=2.0)
sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
ref = oatt.attention_model_forward(sd, x, labels, sl, lsl, L, att, clip_enc=50.0, clip_dec=50.0, sharpening=1.5,
                                   temperature=2.0)
loss, logits, out_train, out_infer = model.compute_loss(x, labels, sl, lsl, 1.0, 1.0, 1.0)
lg = logits.cpu().numpy()
e_logits = np.abs(lg - ref['logits'] * 2.0)
e_alpha = np.abs(out_train.attention_weights.cpu().numpy() - ref['alphas'])
e_loss = abs(loss.item() - ref['total_loss']) / abs(ref['total_loss'])
bad = e_logits.max() > 2e-4 or e_alpha.max() > 1e-5 or e_loss > 1e-4
print('%s %s loss %.2e logits %.2e alpha %.2e' % (tag, 'MISMATCH' if bad else 'ok', e_loss, e_logits.max(), e_alpha.max()), flush=True)
if bad:
    enc = model.encoder._out_tm.cpu().numpy()                 # [T,Bp,2H]
    print('  logits err per step', np.round(e_logits.max(axis=(0, 2)), 6).tolist())
    print('  logits err per row ', np.round(e_logits.max(axis=(1, 2)), 6).tolist())
    print('  alpha  err per step', np.round(e_alpha.max(axis=(0, 2)), 8).tolist())
    print('  encoder out: max |x| %.4f, nan %d; a second forward of the same model:' % (np.abs(enc).max(), int(np.isnan(enc).sum())))
    loss2, logits2, _, _ = model.compute_loss(x, labels, sl, lsl, 1.0, 1.0, 1.0)
    lg2 = logits2.cpu().numpy()
    print('  second run logits err %.2e, first-vs-second %.2e, encoder first-vs-second %.2e' % (
        np.abs(lg2 - ref['logits'] * 2.0).max(), np.abs(lg2 - lg).max(),
        np.abs(model.encoder._out_tm.cpu().numpy() - enc).max()))
    sys.exit(1)

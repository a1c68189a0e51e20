"""TEST INFRASTRUCTURE: one training run of a small fixed configuration in a FRESH process, printed as SHA-256 digests.

    python tests/_determinism_worker.py <key>     ->  one JSON line {"key":..., "step1": {...}, "step2": {...}}

tests/test_gpu_determinism.py starts several of these per key and demands bitwise identical digests: the same seeded
weights, batch and dropout stream must give the same loss, logits and gradient BITS in every process, cold device or not
(a one-off 1.2e-3 logit difference in the first process on a fresh box, never reproduced).  Each
digest covers the raw bytes of the tensor; "grads" is the digest of all gradients in parameter order.  step2 runs from
the parameters step1's update produced, so it also covers the optimizer and every cached weight image being refreshed."""
import hashlib
import json
import sys

import numpy as np
import torch


def digest(t):
    a = t.detach().cpu().contiguous().numpy() if torch.is_tensor(t) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()[:24]


def ctc_run(key):
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
    cfg = {
        'cfgA': dict(enc='blstm', B=16, T=120, D=120, H=128, L=2, C=39, dtype='f32', keep=0.5),
        'headline': dict(enc='blstm', B=16, T=150, D=120, H=256, L=5, C=61, dtype='bf16', keep=0.8),
        'vgg': dict(enc='vgg_blstm', B=20, T=40, D=120, H=512, L=2, C=28, dtype='bf16', keep=0.8, splice=11),
    }[key]
    rng = np.random.RandomState(7)
    B, T, D, C = cfg['B'], cfg['T'], cfg['D'] * cfg.get('splice', 1), cfg['C']
    sl = rng.randint(T // 3, T + 1, size=B).astype(np.int32)
    sl[0] = T
    x = rng.randn(B, T, D).astype(np.float32)
    dense = np.full((B, T // 8), -1, dtype=np.int64)
    for b in range(B):
        x[b, sl[b]:] = 0
        n = max(1, int(sl[b]) // 8)
        dense[b, :n] = rng.randint(0, C, size=n)
    kw = dict(splice=cfg['splice']) if 'splice' in cfg else {}
    model = CTC(encoder_type=cfg['enc'], input_size=cfg['D'], num_units=cfg['H'], num_layers=cfg['L'], num_classes=C,
                parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype=cfg['dtype'], device='cuda:0', seed=3,
                **kw)
    labels = list2sparsetensor(dense, -1)
    out = {}
    opt = None
    for step in (1, 2):
        loss, logits = model.compute_loss(x, labels, sl, keep_prob=cfg['keep'])
        rec = dict(loss=digest(loss), logits=digest(logits))
        opt = opt or model._set_optimizer('rmsprop', 1e-3)
        gv = opt.compute_gradients(loss, model=model)
        rec['grads'] = digest(np.concatenate([g.detach().cpu().numpy().reshape(-1) for g, _ in gv]))
        model._clip_gradients(gv)
        opt.apply_gradients(gv)
        rec['params'] = digest(np.concatenate([model.store[n].detach().cpu().numpy().reshape(-1)
                                               for n in model.store.names]))
        out['step%d' % step] = rec
    return out


def att_run(key):
    from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
    from tensorflow_end2end_speech_recognition_amd.models.attention.joint_ctc_attention import JointCTCAttention
    cfg = {
        'bahdanau': dict(att='bahdanau_content', joint=None, H=64, L=2, U=64, A=32, Em=16, dtype='f32', prev='zeros'),
        'cfgD_toy': dict(att='location', joint=0.5, H=512, L=2, U=512, A=128, Em=64, dtype='bf16', prev='carry'),
    }[key]
    rng = np.random.RandomState(11)
    B, T, D, C, To = 6, 90, 24, 30, 12
    sl = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    sl[0] = T
    x = rng.randn(B, T, D).astype(np.float32)
    lens = rng.randint(3, To - 1, size=B)
    lens[1] = To - 2
    labels = np.full((B, To), C + 1, dtype=np.int64)
    ctc = np.full((B, To - 2), -1, dtype=np.int64)
    for b in range(B):
        x[b, sl[b]:] = 0
        y = rng.randint(0, C, size=lens[b])
        labels[b, 0] = C
        labels[b, 1:1 + lens[b]] = y
        ctc[b, :lens[b]] = y
    kw = dict(input_size=D, encoder_type='blstm', encoder_num_units=cfg['H'], encoder_num_layers=cfg['L'],
              encoder_num_proj=None, attention_type=cfg['att'], attention_dim=cfg['A'], decoder_type='lstm',
              decoder_num_units=cfg['U'], decoder_num_layers=1, embedding_dim=cfg['Em'], num_classes=C, sos_index=C,
              eos_index=C + 1, max_decode_length=To + 3, parameter_init=0.1, clip_grad_norm=5.0,
              clip_activation_encoder=50, clip_activation_decoder=50, dtype=cfg['dtype'], seed=5,
              prev_alpha=cfg['prev'], device='cuda:0')
    model = AttentionSeq2Seq(**kw) if cfg['joint'] is None else JointCTCAttention(lambda_weight=cfg['joint'], **kw)
    out = {}
    opt = None
    for step in (1, 2):
        if cfg['joint'] is None:
            loss, logits, *_ = model.compute_loss(x, labels, sl, lens + 2, 0.8, 0.8, 0.8)
        else:
            loss, logits, *_ = model.compute_loss(x, labels, ctc, sl, lens + 2, 0.8, 0.8, 0.8)
        rec = dict(loss=digest(loss), logits=digest(logits))
        opt = opt or model._set_optimizer('adam', 1e-3)
        gv = opt.compute_gradients(loss, model=model)
        rec['grads'] = digest(np.concatenate([g.detach().cpu().numpy().reshape(-1) for g, _ in gv]))
        model._clip_gradients(gv)
        opt.apply_gradients(gv)
        rec['params'] = digest(np.concatenate([model.store[n].detach().cpu().numpy().reshape(-1)
                                               for n in model.store.names]))
        out['step%d' % step] = rec
    ids = model.infer(x, sl)
    out['infer'] = digest(np.asarray(ids))
    return out


def main():
    key = sys.argv[1]
    import os
    if os.environ.get('ASR_POISON_LDS') == '1':        # what tests/conftest.py does in front of every GPU test
        from tensorflow_end2end_speech_recognition_amd import _lib, ops as _ops
        h = _lib.handle(0, 0)
        h.check(h.lib.asr_debug_poison_lds(h.h, _ops._s()), 'asr_debug_poison_lds')
    out = ctc_run(key) if key in ('cfgA', 'headline', 'vgg') else att_run(key)
    from tensorflow_end2end_speech_recognition_amd import ops
    ops.check_async_errors()
    out['key'] = key
    print('DIGEST ' + json.dumps(out, sort_keys=True))


if __name__ == '__main__':
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()

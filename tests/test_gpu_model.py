"""GPU parity: the whole CTC model (class surface of models/ctc/ctc.py) vs the oracle --
loss within 1e-4 relative (fp32), every parameter gradient, one optimizer step,
bit-exact greedy labels."""
import numpy as np
import pytest
import torch

from oracle import decoders as odec
from oracle import model as omodel
from oracle import optim as oopt

pytestmark = pytest.mark.gpu


def _batch(rng, B, T, D, C, lo=None):
    x = rng.randn(B, T, D).astype(np.float32)
    sl = rng.randint(lo or max(2, T // 2), T + 1, size=B).astype(np.int32)
    sl[0] = T
    labs = []
    for b in range(B):
        x[b, sl[b]:] = 0
        L = max(1, sl[b] // 4)
        labs.append([int(v) for v in rng.randint(0, C, size=L)])
    Lmax = max(len(l) for l in labs)
    dense = np.full((B, Lmax), -1, dtype=np.int64)
    for b, l in enumerate(labs):
        dense[b, :len(l)] = l
    return x, sl, labs, dense


@pytest.mark.parametrize('enc,B,T,D,H,L,C', [('blstm', 16, 40, 120, 128, 2, 39), ('blstm', 5, 23, 12, 64, 3, 10),
                                              ('lstm', 16, 30, 24, 64, 2, 20)])
def test_ctc_model_loss_grads_and_step(cuda, enc, B, T, D, H, L, C):
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
    rng = np.random.RandomState(B + T)
    x, sl, labs, dense = _batch(rng, B, T, D, C)
    model = CTC(encoder_type=enc, input_size=D, num_units=H, num_layers=L, num_classes=C,
                parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, weight_decay=1e-4,
                dtype='f32', device='cuda:0', seed=3)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ref = omodel.ctc_model_forward(sd, x, labs, sl, L, ndir=2 if enc == 'blstm' else 1, cell_clip=50.0,
                                   weight_decay=1e-4)
    loss, logits = model.compute_loss(x, list2sparsetensor(dense, -1), sl, keep_prob=1.0)
    assert logits.shape == (T, B, C + 1)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(model.ctc_losses.cpu().numpy() - ref['ctc_losses']).max() / ref['ctc_losses'].max() < 1e-4
    assert np.abs(logits.cpu().numpy() - ref['logits']).max() < 1e-4
    # greedy labels bit-exact vs the oracle decoder on the oracle logits
    dec = model.decoder(logits, sl, beam_width=1)
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import sparsetensor2list
    hyp = sparsetensor2list(dec, B)
    ref_hyp = odec.greedy_decode(np.transpose(ref['logits'], (1, 0, 2)), sl, C)
    assert [list(h) for h in hyp] == ref_hyp
    ler = model.compute_ler(dec, list2sparsetensor(dense, -1))
    assert 0 <= ler
    dec_b = model.decoder(logits, sl, beam_width=4, merge_repeated=False)       # device prefix beam search
    ref_b, _ = odec.beam_search_decode(__import__('oracle.ctc', fromlist=['x']).log_softmax(np.transpose(ref['logits'], (1, 0, 2))), sl, C, 4)
    assert [list(h) for h in sparsetensor2list(dec_b, B)] == ref_b
    # default = the reference's call, tf.nn.ctc_beam_search_decoder(merge_repeated=True): repeats of the output collapse
    dec_m = [list(h) for h in sparsetensor2list(model.decoder(logits, sl, beam_width=4), B)]
    assert dec_m == [[v for i, v in enumerate(h) if i == 0 or v != h[i - 1]] for h in ref_b]
    # gradients (before clipping)
    opt = model._set_optimizer('momentum', 0.01)
    gv = opt.compute_gradients(loss, model=model)
    worst = 0
    for g, name in gv:
        r = ref['grads'][name]
        rel = np.abs(g.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-8)
        worst = max(worst, rel)
        assert rel < 2e-3, (name, rel)
    # clip + step vs oracle
    model._clip_gradients(gv)
    opt.apply_gradients(gv)
    for name in model.store.names:
        g = oopt.clip_by_norm(ref['grads'][name], 5.0)
        p_ref = sd[name].astype(np.float64) - 0.01 * g       # momentum, first step
        assert np.abs(model.store[name].cpu().numpy() - p_ref).max() < 1e-5, name


def test_ctc_model_train_decreases_loss_and_bf16_close(cuda):
    """overfit one small batch (the reference's own test strategy, models/test/test_ctc.py:225-233)
    and check the bf16 operand path tracks the fp32 path."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(0)
    B, T, D, H, L, C = 16, 50, 120, 128, 2, 39
    x, sl, labs, dense = _batch(rng, B, T, D, C)
    losses = {}
    for dtype in ('f32', 'bf16'):
        model = CTC('blstm', D, H, L, C, parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50,
                    dtype=dtype, seed=1)
        cur = []
        for step in range(30):
            loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
            model.train(loss, 'adam', 1e-3)
            cur.append(loss.item())
        losses[dtype] = cur
        assert cur[-1] < 0.7 * cur[0], cur
    assert abs(losses['bf16'][0] - losses['f32'][0]) / losses['f32'][0] < 2e-2


def test_dropout_path_runs_and_masks(cuda):
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(0)
    B, T, D, H, L, C = 16, 20, 12, 64, 2, 10
    x, sl, labs, dense = _batch(rng, B, T, D, C)
    model = CTC('blstm', D, H, L, C, dtype='f32', seed=1)
    l1, _ = model.compute_loss(x, dense, sl, keep_prob=1.0)
    l2, _ = model.compute_loss(x, dense, sl, keep_prob=0.5)
    model.train(l2, 'sgd', 0.1)
    l3, _ = model.compute_loss(x, dense, sl, keep_prob=0.5, is_training=False)   # eval: no dropout
    assert abs(l1.item() - l2.item()) > 1e-4
    assert np.isfinite(l3.item())


def test_ctc_bottleneck_layer_parity(cuda):
    """bottleneck FC + ReLU between encoder and output layer (models/ctc/ctc.py:201-216): loss, logits and every
    gradient vs the oracle; the dropout / bf16 variants train."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(11)
    B, T, D, H, L, C, BN = 16, 21, 24, 64, 2, 9, 40
    x, sl, labs, dense = _batch(rng, B, T, D, C)
    model = CTC(encoder_type='blstm', input_size=D, num_units=H, num_layers=L, num_classes=C, parameter_init=0.1,
                clip_grad_norm=5.0, clip_activation=50, bottleneck_dim=BN, dtype='f32', seed=6)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    assert sd['bottleneck/weights'].shape == (2 * H, BN) and sd['output/weights'].shape == (BN, C + 1)
    sd['bottleneck/biases'] = (rng.randn(BN) * 0.1).astype(np.float32)
    model.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    ref = omodel.ctc_model_forward(sd, x, labs, sl, L, ndir=2, cell_clip=50.0, bottleneck=True)
    loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(logits.cpu().numpy() - ref['logits']).max() < 1e-4
    opt = model._set_optimizer('sgd', 0.1)
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err)
    for dtype in ('f32', 'bf16'):
        m2 = CTC(encoder_type='blstm', input_size=D, num_units=H, num_layers=L, num_classes=C, clip_grad_norm=5.0,
                 clip_activation=50, bottleneck_dim=BN, dtype=dtype, seed=6)
        l0 = None
        for it in range(20):
            l, _ = m2.compute_loss(x, dense, sl, keep_prob=0.9)
            m2.train(l, 'adam', 3e-3)
            l0 = l.item() if l0 is None else l0
        assert l.item() < 0.9 * l0, dtype


@pytest.mark.parametrize('enc,Lm,Ls,BN,proj', [('multitask_blstm', 3, 2, None, None), ('multitask_blstm', 2, 2, 24, None),
                                               ('multitask_lstm', 3, 1, None, None), ('multitask_blstm', 3, 2, None, 24),
                                               ('multitask_blstm', 2, 1, 24, 40)])
def test_multitask_ctc_parity_and_training(cuda, enc, Lm, Ls, BN, proj):
    """MultitaskCTC (models/ctc/multitask_ctc.py): main head on the top layer, sub head on layer num_layers_sub
    (for multitask_lstm the reference's list alias makes that the top layer too); weighted loss, both logits and
    every gradient vs the oracle; then it trains, decodes and scores both tasks.  proj: lstm_impl='LSTMCell' with num_proj
    (multitask_blstm.py:95): the projected layers on the whole-sequence kernels, the sub head on the projected outputs."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.multitask_ctc import MultitaskCTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
    rng = np.random.RandomState(17)
    B, T, D, H, Cm, Cs, w = 6, 19, 12, 64, 9, 4, 0.7
    x, sl, labs_m, dense_m = _batch(rng, B, T, D, Cm)
    labs_s = [[int(v) for v in rng.randint(0, Cs, size=max(1, int(sl[b]) // 5))] for b in range(B)]   # same utterances
    dense_s = np.full((B, max(len(l) for l in labs_s)), -1, dtype=np.int64)
    for b, l in enumerate(labs_s):
        dense_s[b, :len(l)] = l
    model = MultitaskCTC(encoder_type=enc, input_size=D, num_units=H, num_layers_main=Lm, num_layers_sub=Ls,
                         num_classes_main=Cm, num_classes_sub=Cs, main_task_weight=w, parameter_init=0.1,
                         clip_grad_norm=5.0, clip_activation=50, bottleneck_dim=BN, dtype='f32', seed=9,
                         **(dict(lstm_impl='LSTMCell', num_proj=proj) if proj else {}))
    ndir = 2 if enc == 'multitask_blstm' else 1
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    assert sd['output_sub/weights'].shape == (ndir * (proj or H), Cs + 1)
    for k in sd:                                   # non-zero biases so that the bias paths are exercised
        if k.endswith('/biases'):
            sd[k] = (rng.randn(*sd[k].shape) * 0.05).astype(np.float32)
    model.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    ref = omodel.multitask_ctc_model_forward(sd, x, labs_m, labs_s, sl, Lm, Ls if ndir == 2 else Lm, w, ndir=ndir,
                                             cell_clip=50.0, bottleneck=BN is not None, proj=bool(proj))
    loss, logits_m, logits_s = model.compute_loss(x, dense_m, dense_s, sl, keep_prob=1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(logits_m.cpu().numpy() - ref['logits_main']).max() < 1e-4
    assert np.abs(logits_s.cpu().numpy() - ref['logits_sub']).max() < 1e-4
    assert np.abs(model.ctc_losses_sub.cpu().numpy() - ref['ctc_losses_sub']).max() / ref['ctc_losses_sub'].max() < 1e-4
    opt = model._set_optimizer('sgd', 0.1)
    seen = set()
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        seen.add(name)
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
    assert seen == set(ref['grads'])
    first = last = None
    for it in range(25):
        l, lm_, ls_ = model.compute_loss(x, list2sparsetensor(dense_m, -1), list2sparsetensor(dense_s, -1), sl,
                                         keep_prob=0.9)
        model.train(l, 'adam', 3e-3)
        first = l.item() if first is None else first
        last = l.item()
    assert last < 0.9 * first, (first, last)
    dm, ds = model.decoder(lm_, ls_, sl, beam_width=1)
    ler_m, ler_s = model.compute_ler(dm, ds, list2sparsetensor(dense_m, -1), list2sparsetensor(dense_s, -1))
    assert 0.0 <= ler_m and 0.0 <= ler_s
    pm, ps = model.posteriors(lm_, ls_)
    assert pm.shape == (B * T, Cm + 1) and ps.shape == (B * T, Cs + 1)
    assert abs(float(pm.sum(1).mean()) - 1.0) < 1e-5


def test_vgg_blstm_ctc_parity(cuda):
    """VGG front-end + BLSTM + CTC (BASELINE config C topology, small): loss and every gradient vs the oracle."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(2)
    B, T, F, W, H, L, C = 3, 9, 8, 5, 64, 1, 7          # input_size = 3F = 24, splice 5
    D = F * W * 3
    x, sl, labs, dense = _batch(rng, B, T, D, C, lo=4)
    model = CTC(encoder_type='vgg_blstm', input_size=3 * F, splice=W, num_units=H, num_layers=L, num_classes=C,
                parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype='f32', seed=4)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    # non-zero biases so that the bias paths are exercised
    for k in sd:
        if k.endswith('/bias') or k.endswith('/biases'):
            sd[k] = (rng.randn(*sd[k].shape) * 0.05).astype(np.float32)
    model.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    ref = omodel.ctc_model_forward(sd, x, labs, sl, L, ndir=2, cell_clip=50.0, vgg=(F, W))
    loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(logits.cpu().numpy() - ref['logits']).max() < 2e-4
    opt = model._set_optimizer('sgd', 0.1)
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
    # bf16 operands: the implicit-GEMM convolutions against the im2col + GEMM form (same bf16 operands, so
    # the two agree far inside bf16 rounding) and against the oracle evaluated on the bf16-rounded operands
    import os
    grads = {}
    for mode in ('1', '0'):
        os.environ['ASR_VGG_IMPLICIT'] = mode
        try:
            m3 = CTC(encoder_type='vgg_blstm', input_size=3 * F, splice=W, num_units=H, num_layers=L, num_classes=C,
                     parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=4)
            m3.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
            l3, _ = m3.compute_loss(x, dense, sl, keep_prob=1.0)
            assert abs(l3.item() - ref['total_loss']) / abs(ref['total_loss']) < 2e-2
            opt3 = m3._set_optimizer('sgd', 0.1)
            grads[mode] = {name: g.cpu().numpy().copy() for g, name in opt3.compute_gradients(l3, model=m3)}
        finally:
            os.environ.pop('ASR_VGG_IMPLICIT', None)
    from oracle import lstm as olstm
    ref16 = omodel.ctc_model_forward(sd, x, labs, sl, L, ndir=2, cell_clip=50.0, vgg=(F, W),
                                     operand_round=olstm.bf16_round_t)     # the device path's rounding points
    for name, g in grads['1'].items():
        if 'VGG' in name or 'bridge' in name:
            r = ref16['grads'][name]
            assert np.abs(g - grads['0'][name]).max() < 1e-2 * np.abs(r).max(), name
            assert np.abs(g - r).max() < 2e-2 * np.abs(r).max(), (name, np.abs(g - r).max() / np.abs(r).max())
    # dropout path + bf16 operands run and train
    m2 = CTC(encoder_type='vgg_blstm', input_size=3 * F, splice=W, num_units=H, num_layers=L, num_classes=C,
             clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=4)
    l0 = None
    for it in range(25):
        l, _ = m2.compute_loss(x, dense, sl, keep_prob=0.9)
        m2.train(l, 'adam', 2e-3)
        l0 = l.item() if l0 is None else l0
    assert l.item() < 0.8 * l0


@pytest.mark.parametrize('B', [3, 6])
def test_vgg_dropout_in_the_producing_kernels_equals_the_separate_passes(cuda, B):
    """The bf16 VGG front-end with tf.nn.dropout applied in the convolution / max-pool epilogues (default) against the
    same step with every dropout as its own pass over the stored activation (front.fused_drop = False), same Philox
    counters: identical loss, logits and -- up to the summation order of the atomically accumulated weight gradients --
    every gradient.  B = 3: 42 images (tiled convolution kernels); B = 6: 84 images (image-resident kernels)."""
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(31 + B)
    T, F, W, H, L, C = 14, 40, 11, 256, 1, 28
    x, sl, labs, dense = _batch(rng, B, T, F * W * 3, C, lo=6)
    model = CTC(encoder_type='vgg_blstm', input_size=3 * F, splice=W, num_units=H, num_layers=L, num_classes=C,
                parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=7)
    opt = model._set_optimizer('sgd', 0.1)
    runs = {}
    for fused in (True, False, False):
        model.encoder.front.fused_drop = fused
        calls = model._dropout_calls
        loss, logits = model.compute_loss(x, dense, sl, keep_prob=0.8)
        assert model.encoder.front.ctx['fused_drop'] == fused
        gv = opt.compute_gradients(loss, model=model)
        torch.cuda.synchronize()
        runs.setdefault(fused, []).append((loss.item(), logits.clone(), model.store.grad.clone()))
        model._dropout_calls = calls                      # replay the same masks
    assert ops.check_async_errors(0) == 0
    (lf, zf, gf), = runs[True]
    (l0, z0, g0), (l1, z1, g1) = runs[False]
    assert lf == l0 == l1 and torch.equal(zf, z0)
    scale = float(g0.abs().max())
    noise = float((g1 - g0).abs().max())                  # run-to-run spread of the separate-pass step itself
    diff = float((gf - g0).abs().max())
    print('\nfused vs separate dropout: loss %.6f  grad max |diff| %.3e  (run-to-run %.3e, max |g| %.3e)'
          % (lf, diff, noise, scale))
    assert scale > 0 and diff <= max(4 * noise, 2e-6 * scale)


def test_vgg_blstm_bf16_parity_at_the_cfgC_image_size(cuda):
    """BASELINE configs[2] front-end at its own image size (40 mel bins x splice 11 x {static, delta, delta-delta}, the
    implicit-GEMM convolutions with 64 / 128 channels on MFMA) + one 512-unit BLSTM layer (the 8-CU cluster kernels)
    + CTC, bf16 operands, on three ragged utterances -- against the oracle evaluated with the device path's rounding
    points (inputs, filters, bridge / LSTM / output weights, every stored activation and every emitted h rounded to
    bf16, straight-through; biases, state and all accumulation fp64): loss, logits and EVERY gradient.
    Reference: models/encoders/core/vgg_blstm.py:107-177, cnn_util.py:13-84."""
    from oracle import lstm as olstm
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(12)
    B, T, F, W, H, L, C = 3, 14, 40, 11, 512, 1, 28
    D = F * W * 3
    x, sl, labs, dense = _batch(rng, B, T, D, C, lo=6)
    model = CTC(encoder_type='vgg_blstm', input_size=3 * F, splice=W, num_units=H, num_layers=L, num_classes=C,
                parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=7)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    for k in sd:      # non-zero biases so that the bias paths are exercised
        if k.endswith('/bias') or k.endswith('/biases'):
            sd[k] = (rng.randn(*sd[k].shape) * 0.05).astype(np.float32)
    model.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
    opt = model._set_optimizer('sgd', 0.1)
    gv = opt.compute_gradients(loss, model=model)
    assert ops.check_async_errors(0) == 0
    ref = omodel.ctc_model_forward(sd, x, labs, sl, L, ndir=2, cell_clip=50.0, vgg=(F, W),
                                   operand_round=olstm.bf16_round_t)
    rel = abs(loss.item() - ref['total_loss']) / abs(ref['total_loss'])
    lg = logits.cpu().numpy()
    valid = (np.arange(lg.shape[0])[:, None] < sl[None, :])
    elog = np.abs(lg - ref['logits'])[valid]
    report = ['loss %.6f vs oracle %.6f  rel %.2e   logits max abs %.2e (max |logit| %.2f)'
              % (loss.item(), ref['total_loss'], rel, elog.max(), np.abs(ref['logits']).max())]
    worst = 0.0
    for g, name in gv:
        r = ref['grads'][name]
        e = np.abs(g.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-12)
        report.append('%-44s rel-to-max %.2e' % (name, e))
        worst = max(worst, e)
    print('\n' + '\n'.join(report))
    # measured on MI355X: loss rel 2e-5 .. 6e-5, logits 1.9e-2 .. 2.1e-2 abs (|logit| <= 3.4), gradients 0.7e-2 .. 2.6e-2
    # of their max over the three summation orders the first layer has had (im2col GEMM, lean TN GEMM, direct -- whose
    # outputs agree with each other on 99.998 % of the elements and are all within one bf16 step of the fp64 value,
    # scripts/check_smallc.py): a stored bf16 activation whose fp32-accumulated value lies near a rounding boundary lands
    # one bf16 step (2^-8) away from the fp64-accumulated oracle's, and WHICH ones do moves the worst gradient entry
    # between 1.5e-2 and 2.6e-2
    assert rel < 2e-3, report[0]
    assert elog.max() < 3e-2 * max(1.0, np.abs(ref['logits']).max()), report[0]
    assert worst < 3e-2, '\n'.join(report)


@pytest.fixture
def gru_mode():
    from tensorflow_end2end_speech_recognition_amd import ops

    def set_mode(persistent):
        ops.debug_set_gru_persistent(persistent)
    yield set_mode
    ops.debug_set_gru_persistent(1)


@pytest.mark.parametrize('persistent', [1, 0])
@pytest.mark.parametrize('enc,B,T,D,H,L,C', [('bgru', 16, 37, 24, 64, 2, 12), ('gru', 5, 21, 12, 32, 2, 9),
                                            ('bgru', 20, 90, 42, 256, 1, 30)])
def test_gru_ctc_model_loss_grads_and_step(cuda, gru_mode, enc, B, T, D, H, L, C, persistent):
    """CTC(encoder_type='gru' | 'bgru') on the HIP GRU kernels (csrc/gru.hip) against the oracle's GRU model
    (oracle/gru.py, cell pinned to TensorFlow's testGRUCell): loss 1e-4, logits, EVERY gradient, final states, greedy
    labels bit-exact, ragged lengths incl. a zero-padded batch tile; then training lowers the loss.
    Reference: models/encoders/core/gru.py:9-152, models/ctc/ctc.py:150-155."""
    gru_mode(persistent)     # 1: one persistent launch per layer call (state in LDS, exact-fp32 MFMA); 0: launch per step
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import sparsetensor2list
    rng = np.random.RandomState(B + T)
    x, sl, labs, dense = _batch(rng, B, T, D, C)
    ndir = 2 if enc == 'bgru' else 1
    model = CTC(encoder_type=enc, input_size=D, num_units=H, num_layers=L, num_classes=C, parameter_init=0.2,
                clip_grad_norm=5.0, dtype='f32', seed=3)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    for k in sd:
        if k.endswith('/bias') or k.endswith('/biases'):
            sd[k] = sd[k] + (rng.randn(*sd[k].shape) * 0.05).astype(np.float32)
    model.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    ref = omodel.gru_ctc_model_forward(sd, x, labs, sl, L, ndir=ndir)
    loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(logits.cpu().numpy() - ref['logits']).max() < 2e-4
    fin = model.encoder._finals[-1]
    want = ref['final'] if ndir == 2 else (ref['final'],)
    for d in range(ndir):
        assert np.abs(fin[d, :B].cpu().numpy() - want[d].detach().numpy()).max() < 1e-4
    hyp = sparsetensor2list(model.decoder(logits, sl, 1), B)
    assert [list(h) for h in hyp] == odec.greedy_decode(np.transpose(ref['logits'], (1, 0, 2)), sl, C)
    opt = model._set_optimizer('sgd', 0.1)
    seen = set()
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        seen.add(name)
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
    assert seen == set(ref['grads'])
    l0 = None
    for it in range(12):
        l, _ = model.compute_loss(x, dense, sl, keep_prob=0.9)
        model.train(l, 'adam', 3e-3)
        l0 = l.item() if l0 is None else l0
    assert l.item() < 0.9 * l0


@pytest.mark.parametrize('dtype,proj', [('f32', None), ('bf16', None), ('f32', 40)])
def test_cldnn_ctc_model_parity(cuda, dtype, proj):
    """CTC(encoder_type='cldnn_wang') on the device: the three strided SAME convolutions (asr_im2col + MFMA GEMM with
    fused bias + ReLU, gradients through asr_col2im), the BLSTM stack, fc1 / fc2 -- against the oracle
    (oracle/cldnn.py): loss, logits and every gradient in fp32; the bf16 operand path against the oracle evaluated with
    the device's rounding points (operands, stored activations, emitted h rounded to bf16), and training.  Image 13 mel bins x splice 7 (conv outputs 5x4 -> 5x2 -> 5x2x96).
    Reference: models/encoders/core/cldnn_wang.py:134-249."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.models.encoders.core import cldnn_wang
    rng = np.random.RandomState(21)
    B, T, F, W, H, L, C = 4, 11, 13, 7, 64, 2, 9
    D = F * W * 3
    x, sl, labs, dense = _batch(rng, B, T, D, C, lo=5)
    old_chunk = cldnn_wang.CHUNK_FRAMES
    cldnn_wang.CHUNK_FRAMES = 24                              # several chunks of frames, the last one partial
    try:
        model = CTC(encoder_type='cldnn_wang', input_size=3 * F, splice=W, num_units=H, num_layers=L, num_classes=C,
                    parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype=dtype, seed=4,
                    **(dict(lstm_impl='LSTMCell', num_proj=proj) if proj else {}))    # (cldnn_wang.py:202 passes num_proj on)
        sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
        for k in sd:
            if k.endswith('/bias') or k.endswith('/biases'):
                sd[k] = (rng.randn(*sd[k].shape) * 0.05 + 0.02).astype(np.float32)
        model.store.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
        from oracle import lstm as olstm
        ref = omodel.cldnn_ctc_model_forward(sd, x, labs, sl, L, F, W, cell_clip=50.0, proj=bool(proj),
                                             operand_round=olstm.bf16_round_t if dtype == 'bf16' else None)
        loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
        tol_l, tol_g = (1e-4, 2e-3) if dtype == 'f32' else (2e-3, 5e-2)   # bf16: measured up to 3.2e-2 (one-step bf16 flips of stored activations)
        assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < tol_l
        opt = model._set_optimizer('sgd', 0.1)
        seen = set()
        for g, name in opt.compute_gradients(loss, model=model):
            r = ref['grads'][name]
            seen.add(name)
            err = np.abs(g.cpu().numpy() - r).max()
            assert err < tol_g * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
        assert seen == set(ref['grads'])
        l0 = None
        for it in range(10):
            l, _ = model.compute_loss(x, dense, sl, keep_prob=0.9)
            model.train(l, 'adam', 2e-3)
            l0 = l.item() if l0 is None else l0
        assert l.item() < 0.95 * l0
    finally:
        cldnn_wang.CHUNK_FRAMES = old_chunk


def test_end_to_end_recipe_on_synthetic_corpus(cuda, tmp_path):
    """examples/synthetic/train_ctc.py: dataset iterator -> compute_loss/train -> decoder/compute_ler -> LR
    controller -> Saver, i.e. the call sequence of the reference's train_ctc.py, learns the synthetic corpus and
    its best checkpoint restores into a fresh model."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('synth_train_ctc', os.path.join(root, 'examples', 'synthetic', 'train_ctc.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(['--epochs', '5', '--save_path', str(tmp_path), '--units', '256'])
    hist = out['history']
    assert len(hist) == 5 and out['best'] < 0.35 and out['best'] < 0.6 * hist[0], hist
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver
    fresh = CTC(encoder_type='blstm', input_size=24, num_units=256, num_layers=2, num_classes=12, clip_grad_norm=5.0,
                clip_activation=50, dtype='bf16', seed=99)
    Saver().restore(fresh, out['checkpoint'])
    assert abs(mod.evaluate(fresh, out['dev']) - out['best']) < 1e-6


def test_timit_recipe_on_generated_corpus(cuda, tmp_path):
    """examples/timit/training/train_ctc.py + evaluation/eval_ctc.py on the HIP path (bf16 operands): the run
    trains to a dev PER far below the start, writes the reference's run-directory files, and the evaluation script
    reproduces the test PER from the checkpoint.  (The same flow runs on CPU stand-ins in tests/test_host_logic.py.)"""
    import os
    import sys
    import yaml
    from _corpus import make_timit_like
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from examples.timit.training import train_ctc
    from examples.timit.evaluation import eval_ctc
    corpus = str(tmp_path / 'corpus')
    make_timit_like(corpus, np.random.RandomState(0), n_train=64, n_dev=8, n_test=6, feat=12)
    with open(os.path.join(root, 'examples/timit/config/ctc/blstm_ctc_phone61.yml')) as f:
        cfg = yaml.safe_load(f)
    cfg['param'].update(input_size=12, num_units=64, num_layers=2, batch_size=16, num_epoch=10, eval_start_epoch=1,
                        print_step=4, optimizer='adam', learning_rate=0.01, dropout=0.1, decay_start_epoch=5,
                        dtype='bf16', dataset_root=corpus, sort_stop_epoch=2)
    cfg_path = str(tmp_path / 'cfg.yml')
    with open(cfg_path, 'w') as f:
        yaml.safe_dump(cfg, f)
    import random
    random.seed(0)                    # the iterators draw from the global generator, as the reference's do
    res = train_ctc.main(cfg_path, str(tmp_path / 'runs'))
    run = res['save_path']
    for name in ('config.yml', 'train.log', 'complete.txt', 'loss.csv', 'ler.csv', 'checkpoint'):
        assert os.path.isfile(os.path.join(run, name)), name
    assert len(res['ler_dev']) == 10 and min(res['ler_dev']) < 0.5, res['ler_dev']
    per = eval_ctc.main([run, '--beam_width', '1'])
    assert abs(per - res['ler_test']) < 1e-9
    per_beam = eval_ctc.main([run, '--beam_width', '8'])
    assert 0.0 <= per_beam < 2.0


def test_librispeech_recipe_single_rank(cuda, tmp_path):
    """examples/librispeech/training/train_ctc.py with one rank on the HIP path (the N-rank form of the same driver
    runs under gloo with CPU stand-ins in tests/test_distributed_cpu.py): trains, evaluates CER/WER on both dev
    sets, checkpoints."""
    import os
    import sys
    import yaml
    from _corpus import make_librispeech_like
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from examples.librispeech.training import train_ctc
    corpus = str(tmp_path / 'corpus')
    make_librispeech_like(corpus, np.random.RandomState(1), n_train=48, n_other=6, feat=12)
    with open(os.path.join(root, 'examples/librispeech/config/ctc/blstm_ctc_character_100h.yml')) as f:
        cfg = yaml.safe_load(f)
    cfg['param'].update(input_size=12, num_stack=1, num_skip=1, num_units=64, num_layers=2, batch_size=16, num_epoch=8,
                        eval_start_epoch=1, print_step=3, learning_rate=0.01, dropout=0.1, dataset_root=corpus,
                        sort_stop_epoch=2, decay_start_epoch=4)
    cfg_path = str(tmp_path / 'cfg.yml')
    with open(cfg_path, 'w') as f:
        yaml.safe_dump(cfg, f)
    import random
    random.seed(0)                    # the iterators draw from the global generator, as the reference's do
    res = train_ctc.main(cfg_path, str(tmp_path / 'runs'))
    assert res['world'] == 1 and res['steps'] == 24 and len(res['metric_dev']) == 8
    assert min(res['metric_dev']) < 0.6 * res['metric_dev'][0], res['metric_dev']
    assert res['checkpoints'] and res['test'] is not None
    for name in ('config.yml', 'train.log', 'complete.txt', 'loss_ler.csv', 'checkpoint'):
        assert os.path.isfile(os.path.join(res['save_path'], name)), name


def test_rccl_collectives_on_the_parameter_store(cuda):
    """The RCCL calls of the data-parallel step (utils/training/multi_gpu.py: init with device_id, broadcast of the
    flat parameter buffer, all-reduce + scale of the flat gradient buffer, scalar mean) issued for real on the GPU.
    The box has one GPU, so the group has one rank: this checks API use, dtypes, contiguity and stream ordering
    against the HIP kernels around them -- the N-rank arithmetic is covered by the gloo tests."""
    import os
    import socket
    import torch.distributed as dist
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=cuda)
    except Exception as e:                                   # no RCCL in this environment: nothing to check
        pytest.skip('RCCL process group could not be created: %r' % (e,))
    try:
        rng = np.random.RandomState(0)
        B, T, D, H, C = 16, 20, 12, 64, 7
        x, sl, labs, dense = _batch(rng, B, T, D, C)
        model = CTC('blstm', D, H, 1, C, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=1)
        before = model.store.flat.clone()
        dist.broadcast(model.store.flat, src=0)
        assert torch.equal(before, model.store.flat)
        opt = model._set_optimizer('adam', 1e-3)
        loss, _ = model.compute_loss(x, dense, sl, keep_prob=1.0)
        gv = opt.compute_gradients(loss, model=model)
        model._clip_gradients(gv)
        g = model.store.grad.clone()
        dist.all_reduce(model.store.grad, op=dist.ReduceOp.SUM)         # what average_gradients issues for N > 1
        assert torch.equal(g, model.store.grad)
        t = multi_gpu.average_scalar(loss.detach())
        assert abs(float(t) - float(loss)) < 1e-6
        opt.apply_gradients(gv)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.isfinite(model.store.flat).all()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('bucket_mb', ['0', '1000'])
def test_bucketed_gradient_averaging_on_the_communication_stream(cuda, monkeypatch, bucket_mb):
    """multi_gpu.BucketedAverager on the device: per-layer clip (+ collective) on a communication stream hung on the
    layers' gradient events while the BPTT kernels of the layers below run, the rest after the backward pass -- the
    clipped gradient buffer must equal the single-bucket path bit for bit, for the fp32 single-CU kernels and for the
    bf16 cluster kernels (one rank: the tower mean is the identity; the N-rank arithmetic is covered by the gloo
    tests, the collective itself by the RCCL tests below)."""
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu
    rng = np.random.RandomState(4)
    monkeypatch.setenv('ASR_DP_BUCKET_MB', bucket_mb)   # 0: one collective per layer; 1000: all layers share one
    for dtype, H, L in (('f32', 64, 3), ('bf16', 256, 3)):
        B, T, D, C = 16, 60, 24, 9
        x, sl, labs, dense = _batch(rng, B, T, D, C)
        model = CTC('blstm', D, H, L, C, parameter_init=0.3, clip_grad_norm=0.01, clip_activation=50, dtype=dtype, seed=2)
        opt = model._set_optimizer('sgd', 0.1)
        loss, _ = model.compute_loss(x, dense, sl, keep_prob=0.9)
        gv = opt.compute_gradients(loss, model=model)
        model._clip_gradients(gv)
        want = model.store.grad.clone()
        model._dropout_calls -= 1                       # replay the same dropout masks
        avg = multi_gpu.averager_for(model)
        assert avg.ok and len(avg.buckets) == (L if bucket_mb == '0' else 1)
        avg.force = True
        loss2, _ = model.compute_loss(x, dense, sl, keep_prob=0.9)
        multi_gpu.clip_and_average(model, opt, loss2)
        torch.cuda.synchronize()
        assert abs(loss2.item() - loss.item()) == 0.0
        assert torch.equal(model.store.grad, want), dtype
        opt.apply_gradients(None)
        assert ops.check_async_errors(0) == 0


def test_native_rccl_allreduce_mean_one_rank(cuda):
    """asr_comm_unique_id / asr_comm_init / asr_allreduce_mean (the collective of the C ABI, RCCL opened with dlopen)
    with a one-rank communicator on the box's single GPU: library binding, id bootstrap, in-place all-reduce on the
    launch stream ordered against HIP kernels before and after it; value unchanged for world = 1."""
    from tensorflow_end2end_speech_recognition_amd import ops
    comm = ops.NativeComm(0, 0, 1, ops.NativeComm.unique_id())
    x = torch.arange(1, 100004, dtype=torch.float32, device=cuda)        # odd length: scalar tail of the scale kernel
    y = ops.scale_(x.clone(), 3.0)
    comm.allreduce_mean(y)
    y = ops.scale_(y, 0.5)
    assert torch.equal(y.cpu(), (x * 3.0 * 0.5).cpu())
    comm.close()


def _rccl_world2_worker(rank, port, out):
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2',
                      LOCAL_RANK=str(rank))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    from tensorflow_end2end_speech_recognition_amd.utils.parameter import ParamStore
    from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    multi_gpu.init_process_group(dev)
    st = ParamStore(dev)
    st.declare('w', (1000, 37), np.zeros((1000, 37)))
    st.finalize()
    st.grad.copy_(torch.arange(st.grad.numel(), dtype=torch.float32, device=dev) * (rank + 1))
    assert multi_gpu.native_comm(dev) is not None
    multi_gpu.average_gradients(st)
    want = torch.arange(st.grad.numel(), dtype=torch.float32) * 1.5
    out.put((rank, bool(torch.equal(st.grad.cpu(), want))))
    dist.destroy_process_group()


def test_allreduce_scaling_pass_takes_any_float_aligned_buffer(cuda):
    """The x 1/N pass of asr_allreduce_mean (csrc/comm.hip: 16-byte vectors between a scalar head and tail) on buffers that
    start 0..3 floats past a 16-byte boundary, lengths 0 mod 4 and not, down to shorter than the head -- an external C-ABI
    caller need not align.  It only runs with world > 1, hence the test hook."""
    import ctypes
    from tensorflow_end2end_speech_recognition_amd import _lib, ops
    lib = _lib.handle(0).lib
    lib.asr_debug_comm_scale.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p]
    base = torch.arange(1, 5001, dtype=torch.float32, device=cuda)
    for off in range(4):
        for n in (1, 2, 3, 4, 5, 7, 8, 1023, 1024, 4093):
            buf = base.clone()
            view = buf[off:off + n]
            assert lib.asr_debug_comm_scale(ctypes.c_void_p(view.data_ptr()), n, 0.5, ops._s()) == 0
            torch.cuda.synchronize()
            want = base.clone()
            want[off:off + n] *= 0.5
            assert torch.equal(buf, want), (off, n)


def test_native_rccl_allreduce_mean_world2(cuda):
    """Two ranks on two GPUs over xGMI through the C ABI communicator (skipped on a 1-GPU box: the driver's
    multi-GPU node is where this runs)."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_world2_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_bench_line_of_a_two_rank_run(cuda, tmp_path):
    """bench.py under the driver's launcher contract with TWO ranks (`python -m torch.distributed.run --nproc-per-node 2
    bench.py --gpus 2 ...`).  On a 1-GPU box both ranks share cuda:0 and gloo stands in for RCCL (which refuses two
    ranks on one device; ASR_BENCH_DEVICE / ASR_BENCH_BACKEND are dry-run knobs the driver never sets); with two GPUs it
    is the real RCCL run.  Checked: the rank-0 JSON line is a 2-GPU line (n_gpus, parallelism, global batch, frames of
    BOTH ranks), every rank padded to the global Tmax, per-rank medians and the communication stream's collectives are
    reported, the coalesced buckets partition the gradient buffer, loss finite, hand-off error word 0."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    if torch.cuda.device_count() < 2:
        env.update(ASR_BENCH_DEVICE='0', ASR_BENCH_BACKEND='gloo', ASR_DP_COLLECTIVE='torch')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4',
           '--warmup', '2', '--no-parity', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 6000                                   # the driver keeps 8 000 bytes of stdout
    d = json.loads(r.stdout.strip().splitlines()[-1])
    full = json.load(open(os.path.join(root, d['full'])))         # prose, bucket tables: the file, not the line
    assert full['n_gpus'] == 2 and abs(full['value'] - d['value']) < 1e-4 * d['value']
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 32
    assert d['scaling'] == 'weak' and d['steps'] == 4 and d['warmup'] == 2 and d['unit'] == 'frames/s'
    pr = d['per_rank']
    assert len(pr['frames']) == 2 and abs(sum(pr['frames']) - d['config']['frames_per_step']) < 0.5
    assert pr['frames'][0] != pr['frames'][1]                     # every rank has its own shard of the global batch
    assert abs(d['value'] - d['config']['frames_per_step'] * 4 / (d['ms_per_step'] * 4e-3)) < 1e-3 * d['value']
    assert 'global Tmax' in d['config']['padded_to']
    assert d['comm']['allreduce_ms_per_step'] > 0
    comm = full['comm']
    assert comm['allreduce_calls_per_step'] == len(comm['buckets']) and comm['allreduce_ms_per_step'] > 0
    assert abs(sum(b['mbytes'] for b in comm['buckets']) * 1e6 - comm['bytes_per_step']) < 1.0
    assert np.isfinite(d['final_loss']) and d['cluster_handoff_flags'] == 0
    assert 'cfgA' not in d and 'cfgC' not in d                    # auxiliary configurations are N = 1 entries


def test_bench_bare_gpus_2_on_the_device(cuda):
    """`python bench.py --gpus 2 --steps 3 --warmup 1` with no launcher environment (what a scaling driver may run
    verbatim): bench.py starts the two ranks itself and the single stdout line is a measured 2-rank line with per-rank
    medians, the collectives' time, the rank count of the collective and the same shards under the OTHER padding rule
    beside the headline (global Tmax, utils/dataset/ctc.py:171-182).  One GPU: both ranks on cuda:0 over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    if torch.cuda.device_count() < 2:
        env.update(ASR_BENCH_DEVICE='0', ASR_BENCH_BACKEND='gloo', ASR_DP_COLLECTIVE='torch')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--no-parity', '--own-tmax-steps', '2'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6000, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['steps'] == 3 and d['config']['parallelism'] == 'dp2'
    assert 'dry_run' not in d and len(d['per_rank']['step_median_ms']) == 2
    assert d['comm']['ranks'] == 2 and d['comm']['allreduce_ms_per_step'] > 0 and d['comm']['backend'] in ('gloo', 'nccl')
    assert 'global Tmax' in d['config']['padded_to'] and d['other_padding']['padded_to'] == 'own Tmax per rank'
    assert d['other_padding']['value'] > 0 and len(d['other_padding']['per_rank_step_median_ms']) == 2
    assert d['cluster_handoff_flags'] == 0


def test_bench_last_stdout_line_is_the_compact_record(cuda):
    """`python bench.py --steps 2 --warmup 1` as the driver runs it at N = 1: the LAST stdout line parses, fits the
    driver's 8 000-byte window with margin and carries the contract keys + roofline + cpu_baseline; the full object it
    points to exists (a 20.7 KB line lost its head in the driver's record)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--aux', 'decode,D39',
                        '--aux-steps', '1', '--aux-warmup', '1', '--cpu-tmax', '64', '--cpu-threads', '8'],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 6000, len(last)
    d = json.loads(last)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['steps'] == 2 and d['warmup'] == 1 and d['n_gpus'] == 1 and d['dtype'] == 'bf16'
    rf, cb = d['roofline'], d['cpu_baseline']
    assert rf['bound'] in ('hbm', 'mfma') and rf['peak'] > 0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 * rf['frac']
    assert cb['value'] > 0 and cb['cores'] >= 1 and cb['kind'] in ('port', 'reference') and cb['sample']
    assert d['cfgA']['value'] > 0 and d['decode'] and d['input_width_D39']['value'] > 0
    full = json.load(open(os.path.join(root, d['full'])))
    assert abs(full['value'] - d['value']) < 1e-4 * d['value'] and 'note' in full['roofline']


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_overfit_one_utterance_to_low_ler(cuda, dtype):
    """The reference's own model test (models/test/test_ctc.py:170-233): one utterance repeated B = 4 times, adam,
    stop when the label error rate of the greedy decode is below 0.1; the reference allows 1000 steps."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
    rng = np.random.RandomState(0)
    B, T, D, C = 4, 60, 24, 20
    x = np.repeat(rng.randn(1, T, D).astype(np.float32), B, 0)
    sl = np.array([T] * B, np.int32)
    lab = rng.randint(0, C, size=14)
    st = list2sparsetensor(np.repeat(lab[None], B, 0), -1)
    model = CTC('blstm', D, 128, 2, C, parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype=dtype,
                seed=1)
    ler = 1.0
    for step in range(1000):
        loss, logits = model.compute_loss(x, st, sl, 0.9)
        model.train(loss, 'adam', 1e-3)
        if (step + 1) % 20 == 0:
            _, lg = model.compute_loss(x, st, sl, 1.0, is_training=False)
            ler = model.compute_ler(model.decoder(lg, sl, 1), st)
            if ler < 0.1:
                break
    assert ler < 0.1, (step, ler)


def test_cfgB_bf16_model_parity_at_the_benchmarked_shape(cuda):
    """BASELINE configs[1] exactly as bench.py times it (TIMIT-61, 5x256 BLSTM-CTC, bf16 operands, B = 16, D = 120,
    C = 62, seq_len ~ U{100..778}; same seeded batch and the same constructor arguments as bench.py) against the oracle
    evaluated on the bf16-rounded operands (inputs, LSTM kernels, output weights, every emitted / fed-back h rounded,
    straight-through; state and accumulation fp64): mean loss, per-utterance losses, logits, EVERY parameter
    gradient, and the greedy labels (as a label error rate against the oracle's decode: one flipped argmax anywhere
    in 6.4k frames changes a label, so this is a rate, not bit-exactness -- bit-exact decode is asserted on the fp32
    path).  Runs the multi-CU cluster kernels for 778 steps; the hand-off error word must stay 0.
    Reference: models/ctc/ctc.py:175-323, models/encoders/core/blstm.py:258-332."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_batch
    from oracle import lstm as olstm
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import sparsetensor2list
    from tensorflow_end2end_speech_recognition_amd.utils.evaluation.edit_distance import compute_ler
    B, D, H, L, C = 16, 120, 256, 5, 62
    x, sl, labs, dense = make_batch(1, B, D, C, 100, 778)
    model = CTC('blstm', D, H, L, C - 1, parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype='bf16',
                device='cuda:0', seed=0)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    loss, logits = model.compute_loss(x, dense, sl, keep_prob=1.0)
    opt = model._set_optimizer('rmsprop', 1e-3)
    gv = opt.compute_gradients(loss, model=model)
    assert ops.check_async_errors(0) == 0
    ref = omodel.ctc_model_forward(sd, x, labs, sl, L, cell_clip=50.0, operand_round=olstm.bf16_round_t)
    rel = abs(loss.item() - ref['total_loss']) / abs(ref['total_loss'])
    per_utt = np.abs(model.ctc_losses.cpu().numpy() - ref['ctc_losses']).max() / ref['ctc_losses'].max()
    lg = logits.cpu().numpy()
    valid = (np.arange(lg.shape[0])[:, None] < sl[None, :])
    elog = np.abs(lg - ref['logits'])[valid]
    report = ['loss %.6f vs oracle %.6f  rel %.2e   per-utterance rel %.2e   logits max abs %.2e mean abs %.2e'
              % (loss.item(), ref['total_loss'], rel, per_utt, elog.max(), elog.mean())]
    worst = 0.0
    for g, name in gv:
        r = ref['grads'][name]
        e = np.abs(g.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-12)
        report.append('%-44s rel-to-max %.2e' % (name, e))
        worst = max(worst, e)
    hyp = [list(h) for h in sparsetensor2list(model.decoder(logits, sl, beam_width=1), B)]
    ref_hyp = odec.greedy_decode(np.transpose(ref['logits'], (1, 0, 2)), sl, C - 1)
    ler = compute_ler(hyp, ref_hyp)
    report.append('greedy decode vs oracle decode: label error rate %.4f (%d of %d utterances identical)'
                  % (ler, sum(h == r for h, r in zip(hyp, ref_hyp)), B))
    print('\n' + '\n'.join(report))
    assert rel < 2e-3 and per_utt < 2e-3, report[0]
    assert elog.max() < 5e-2 and elog.mean() < 2e-3, report[0]
    assert worst < 2e-2, '\n'.join(report)
    assert ler < 0.02, report[-1]


@pytest.mark.parametrize('ndir,B,T,D,H,P,L,fused', [
    (2, 6, 40, 24, 128, 48, 2, '1'), (1, 20, 25, 12, 64, 32, 2, '1'), (2, 6, 40, 24, 128, 48, 2, '0'),
    (1, 20, 25, 12, 64, 32, 2, '0'), (2, 19, 33, 24, 256, 50, 2, '1'), (2, 16, 21, 21, 320, 128, 1, '1'),
    (2, 7, 17, 12, 192, 21, 2, '1')])
def test_lstmcell_projection_layers(cuda, monkeypatch, ndir, B, T, D, H, P, L, fused):
    """lstm_impl='LSTMCell' with num_proj on the HIP path, both forms of rnn_util.LSTMPLayer -- fused '1': the
    whole-sequence recurrence kernels (clusters at 128 / 256 / 320) on W_p W_h with asr_lstm_bwd_ex's gradient-blocking
    clip and every other product batched over T (batches that do not fill a 16-utterance tile, projection widths that
    are not a multiple of 4); '0': step by step on the generic kernels (skinny MFMA products, asr_lstm_cell_fwd / _bwd).
    Loss 1e-4, logits, every gradient incl. projection/kernel against the oracle's projected-cell model; trains.
    Reference: blstm.py:187-230."""
    import _config_parity as cp
    monkeypatch.setenv('ASR_LSTMP_FUSED', fused)
    r = cp.run_lstmp('cuda:0', B=B, T=T, D=D, H=H, P=P, L=L, C=9, ndir=ndir)
    assert r['trained']
    if ndir == 2:
        print('\n' + r['report'])
        assert r['loss_rel'] < 1e-4 and r['logits_abs'] < 2e-4 and r['grad_worst'] < 2e-3, r['report']
        # an active cell clip (LSTMCell clamps with tf.clip_by_value: a clamped state passes no gradient back)
        rc = cp.run_lstmp('cuda:0', B=B, T=T, D=D, H=H, P=P, L=L, C=9, ndir=2, clip=0.15)
        assert rc['loss_rel'] < 1e-4 and rc['grad_worst'] < 2e-3, rc['report']


@pytest.mark.parametrize('ndir,B,T,D,H,P,clip', [(2, 21, 30, 18, 128, 40, 0.4), (1, 16, 19, 12, 256, 64, None)])
def test_projected_layer_fused_against_step_by_step_with_final_state_gradients(cuda, monkeypatch, ndir, B, T, D, H, P, clip):
    """One rnn_util.LSTMPLayer, its two forms side by side on the same variables: outputs, final (c, m) states, dx and every
    weight gradient -- with gradients arriving through the FINAL states as well (d_final; the CTC models never send one,
    an attention bridge over projected cells would), an active gradient-blocking clip, a batch that does not fill its
    last 16-utterance tile, a dropout mask on the output.  (Weights of +-0.2: at +-0.5 and 256 units the recurrence is
    chaotic -- the two forms, which differ by fp32 rounding only, drift apart by a factor of ten every five steps, measured
    5e-7 at step 1 and 5e-2 at step 30 -- and the comparison would say nothing.)"""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd import ops
    m = CTC(encoder_type='blstm' if ndir == 2 else 'lstm', input_size=D, num_units=H, num_layers=1, num_classes=7,
            lstm_impl='LSTMCell', num_proj=P, parameter_init=0.2, clip_activation=clip, seed=3, dtype='f32', device='cuda:0')
    layer = m.encoder.layers[0]
    st = layer.store
    rng = np.random.RandomState(B + H)
    sl_np = rng.randint(1, T + 1, size=B).astype(np.int32)
    sl_np[0], sl_np[-1] = T, 0
    x = torch.tensor(rng.randn(T, B, D) * 1.5, dtype=torch.float32, device=cuda)
    sl = torch.tensor(sl_np, device=cuda)
    mask = torch.tensor((rng.rand(T, B, ndir * P) < 0.8) / 0.8, dtype=torch.float32, device=cuda)
    dout = torch.tensor(rng.randn(T, B, ndir * P), dtype=torch.float32, device=cuda)
    dfin = [(torch.tensor(rng.randn(B, H) * 0.5, dtype=torch.float32, device=cuda),
             torch.tensor(rng.randn(B, P) * 0.5, dtype=torch.float32, device=cuda)) for _ in range(ndir)]
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('ASR_LSTMP_FUSED', mode)
        assert layer.fused() == (mode == '1')
        out, fin = layer.forward(x, sl, mask)
        dx = layer.backward(dout, dfin, need_dx=True)
        ops.join_side(cuda)
        torch.cuda.synchronize()
        res[mode] = dict(out=out.clone(), dx=dx.clone(), fin=[(c.clone(), mm.clone()) for c, mm in fin],
                         g={n: st.g(n).clone() for n in layer.var_names()})
    def rel(a, b):
        return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
    f, s_ = res['1'], res['0']
    assert rel(f['out'], s_['out']) < 2e-5 and rel(f['dx'], s_['dx']) < 2e-4
    for d in range(ndir):
        assert rel(f['fin'][d][0], s_['fin'][d][0]) < 2e-5 and rel(f['fin'][d][1], s_['fin'][d][1]) < 2e-5
    worst = max((rel(f['g'][n], s_['g'][n]), n) for n in f['g'])
    assert worst[0] < 2e-4, worst
    if clip:     # the clip was active
        monkeypatch.setenv('ASR_LSTMP_FUSED', '1')
        layer.forward(x, sl, None)
        cs = layer.ctx['cs'][:, :B]
        live = (torch.arange(T, device=cuda).unsqueeze(1) < sl.unsqueeze(0)).unsqueeze(2)
        frac = float(((cs.abs() >= clip) & live).float().sum() / (live.float().sum() * cs.shape[2]))
        assert 0.02 < frac < 0.9, frac
        layer.ctx = None
    assert float(f['out'][:, -1].abs().max()) == 0 and float(f['dx'][:, -1].abs().max()) == 0   # the empty utterance


@pytest.mark.parametrize('ndir,B,T,D,H,P,L', [(2, 16, 60, 24, 256, 128, 2), (2, 9, 41, 24, 128, 48, 2), (1, 16, 33, 24, 512, 256, 1),
                                              (2, 16, 27, 24, 320, 160, 1)])
def test_lstmcell_projection_layers_bf16_operands(cuda, monkeypatch, ndir, B, T, D, H, P, L):
    """A bf16 model's projected layers: the bf16 recurrence kernels (the headline's clusters at 256 / 512 / 320, with the
    gradient-blocking clip as their own instantiation) on W_p W_h rounded once, bf16 operands in every batched product, fp32 at
    the layer boundaries.  Against the fp32 oracle at the bars the bf16 BLSTM models are held to (loss 2e-3, gradients a few
    per cent of the largest entry), with an inactive and an ACTIVE clip; and it is the bf16 path that ran.  Weights of +-0.06:
    at the +-0.2 of the fp32 test a 256-unit projected layer amplifies a perturbation tenfold every few steps (fp32 rounding
    survives 60 steps of that, bf16 rounding does not: logits 0.4 apart) and the comparison would measure the dynamics."""
    import _config_parity as cp
    from tensorflow_end2end_speech_recognition_amd.models.encoders.core import rnn_util
    from tensorflow_end2end_speech_recognition_amd._lib import ASR_BF16
    monkeypatch.setenv('ASR_LSTMP_FUSED', '1')      # (the test is about this path: whatever the environment selects)
    monkeypatch.setenv('ASR_LSTMP_BF16', '1')
    seen = []
    orig = rnn_util.LSTMPLayer.operand_dtype
    monkeypatch.setattr(rnn_util.LSTMPLayer, 'operand_dtype', lambda self: seen.append(orig(self)) or seen[-1])
    r = cp.run_lstmp('cuda:0', B=B, T=T, D=D, H=H, P=P, L=L, C=9, ndir=ndir, dtype='bf16', init=0.06)
    assert seen and all(d == ASR_BF16 for d in seen), seen
    if ndir == 2:
        print('\n' + r['report'])
        assert r['loss_rel'] < 2e-3 and r['logits_abs'] < 5e-2 and r['grad_worst'] < 6e-2, r['report']
        rc = cp.run_lstmp('cuda:0', B=B, T=T, D=D, H=H, P=P, L=L, C=9, ndir=2, clip=0.05, dtype='bf16', init=0.06)
        # (an active clip is a discontinuity: a state within bf16 rounding of the threshold is clamped in one run and not in the
        # other, and the elements it feeds differ by their whole contribution -- 0.16 of the largest entry measured on one
        # kernel gradient, 5e-2 in L2 -- while the loss moves by 7e-6)
        assert rc['loss_rel'] < 2e-3 and rc['grad_worst'] < 0.3, rc['report']
    # the A/B switch keeps fp32 operands
    monkeypatch.setenv('ASR_LSTMP_BF16', '0')
    del seen[:]
    r32 = cp.run_lstmp('cuda:0', B=B, T=T, D=D, H=H, P=P, L=L, C=9, ndir=ndir, dtype='bf16', init=0.06)
    assert seen and all(d != ASR_BF16 for d in seen)
    if ndir == 2:
        assert r32['loss_rel'] < 1e-4 and r32['grad_worst'] < 2e-3, r32['report']


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_vgg_front_end_over_projected_cells(cuda, dtype):
    """CTC(encoder_type='vgg_blstm', lstm_impl='LSTMCell', num_proj): the reference hands num_proj of the VGG encoders to the
    same cell builder (models/encoders/core/vgg_blstm.py:107-190 -> blstm.py:187-230).  The front-end in the model's operand
    dtype, the projected stack behind it on the whole-sequence kernels: loss, logits and every gradient (filters, bridge,
    kernels, projection/kernel) against the oracle's composition of the two; fp32 at the fp32 bars, bf16 at the bf16 models'."""
    import _config_parity as cp
    r = cp.run_vgg_lstmp('cuda:0', B=6, T=24, F=16, W=5, H=128, P=48, L=2, C=9, dtype=dtype, init=0.1 if dtype == 'f32' else 0.06)
    print('\n' + r['report'])
    assert r['finite']
    if dtype == 'f32':
        assert r['loss_rel'] < 1e-4 and r['logits_abs'] < 5e-4 and r['grad_worst'] < 5e-3, r['report']
    else:
        assert r['loss_rel'] < 2e-3 and r['logits_abs'] < 5e-2 and r['grad_worst'] < 0.12, r['report']


def test_gru_long_run_keeps_side_lane_bounded_and_survives_poisoned_allocator(cuda):
    """Two regressions of the GRU encoders found in review: (1) the model's head gradients are issued on side lane 1 and
    must be joined (ordered before clip / update, and the lane's keep list released) by the encoder's backward -- 60
    bgru-CTC steps, the keep list stays empty after every step; (2) frames past an utterance's length are skipped by the
    kernels, so whatever they leave there must not reach a weight gradient: the allocator is poisoned with NaNs, the
    batch is ragged, every gradient stays finite and equal to the run on a clean allocator."""
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(5)
    B, T, D, H, C = 6, 33, 12, 32, 7
    x, sl, labs, dense = _batch(rng, B, T, D, C, lo=5)
    sl[1], sl[4] = 7, 12

    def grads(poison):
        if poison:
            junk = [torch.full((1 << 20,), float('nan'), device=cuda) for _ in range(24)]
            del junk                                          # back to the caching allocator, NaN patterns intact
        model = CTC('bgru', D, H, 2, C, parameter_init=0.1, clip_grad_norm=5.0, seed=3)
        loss, _ = model.compute_loss(x, dense, sl, keep_prob=1.0)
        opt = model._set_optimizer('sgd', 0.1)
        return {n: g.cpu().numpy().copy() for g, n in opt.compute_gradients(loss, model=model)}, model
    clean, _ = grads(False)
    dirty, model = grads(True)
    for n in clean:
        assert np.isfinite(dirty[n]).all(), n
        assert np.array_equal(clean[n], dirty[n]), n
    for step in range(60):
        loss, _ = model.compute_loss(x, dense, sl, keep_prob=0.9)
        model.train(loss, 'adam', 1e-3)
        assert all(len(st['keep']) == 0 for st in ops._side.values()), step
    assert np.isfinite(loss.item())


def test_infeasible_labels_raise_without_a_per_step_sync(cuda):
    """tf.nn.ctc_loss(ignore_longer_outputs_than_inputs=False) fails a step whose labels do not fit the frames
    (ctc.py:289).  Here the device counter is watched asynchronously while training (ops.DeferredCheck): the ValueError
    arrives within DEPTH + 1 steps, a feasible run never raises, and the evaluation form raises at once."""
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(0)
    B, T, D, C = 4, 12, 12, 6
    x = rng.randn(B, T, D).astype(np.float32)
    sl = np.array([12, 10, 3, 12], np.int32)
    good = np.full((B, 4), -1, np.int64)
    good[:, :2] = rng.randint(0, C, size=(B, 2))
    bad = good.copy()
    bad[2] = [1, 1, 1, 1]                        # 4 repeats need 7 frames, the utterance has 3
    model = CTC('blstm', D, 64, 1, C, parameter_init=0.1, dtype='f32', seed=0)
    ops.flush_deferred_checks()
    for _ in range(6):
        loss, _ = model.compute_loss(x, good, sl, 1.0)
        model.train(loss, 'sgd', 1e-3)
    ops.flush_deferred_checks()
    with pytest.raises(ValueError, match='Not enough time'):
        for _ in range(ops.DeferredCheck.DEPTH + 2):
            loss, _ = model.compute_loss(x, bad, sl, 1.0)
            model.train(loss, 'sgd', 1e-3)
        ops.flush_deferred_checks()
    ops.flush_deferred_checks()                  # the watch is empty again after it has reported
    with pytest.raises(ValueError, match='Not enough time'):
        model.compute_loss(x, bad, sl, 1.0, is_training=False)
    loss, _ = model.compute_loss(x, good, sl, 1.0, is_training=False)
    assert np.isfinite(float(loss.item()))


def test_deferred_check_depth_is_in_steps_and_one_incident_raises_once(cuda):
    """a model with two CTC heads arms two counters per step; the watch's depth is counted in optimizer STEPS
    (note_step), so two heads do not halve the distance the issue loop may run ahead.  An error drops the other pending
    copies (documented: one exception per incident), and without optimizer steps the pending copies are capped by the
    ring."""
    from tensorflow_end2end_speech_recognition_amd import ops
    ops.flush_deferred_checks()
    w = ops.DeferredCheck(0)
    zero = torch.zeros(1, dtype=torch.int32, device='cuda:0')
    three = torch.full((1,), 3, dtype=torch.int32, device='cuda:0')
    five = torch.full((1,), 5, dtype=torch.int32, device='cuda:0')

    def mk(tag):
        return lambda n: ValueError('%s %d' % (tag, n))
    # two heads per step, all clean: nothing raises, nothing older than DEPTH steps stays pending, and more than DEPTH
    # copies MAY be pending (the old form blocked at DEPTH arm() calls = DEPTH / 2 steps)
    most = 0
    for step in range(3 * ops.DeferredCheck.DEPTH):
        w.arm(zero, mk('main'))
        w.arm(zero, mk('sub'))
        w.note_step()
        most = max(most, len(w.slots))
        assert all(w.step - s[3] <= ops.DeferredCheck.DEPTH for s in w.slots)
        assert len(w.slots) <= 2 * (ops.DeferredCheck.DEPTH + 1)
    w.flush()
    assert not w.slots
    # both heads fail in one step: ONE exception (the older copy's), the rest is dropped
    with pytest.raises(ValueError, match='main 3'):
        w.arm(three, mk('main'))
        w.arm(five, mk('sub'))
        w.note_step()
        w.flush()
    assert not w.slots
    w.flush()
    # no optimizer steps at all (a loop of compute_loss calls): pending copies are capped by the ring
    for _ in range(3 * ops.DeferredCheck.RING):
        w.arm(zero, mk('x'))
        assert len(w.slots) < ops.DeferredCheck.RING
    w.flush()


def test_full_chip_clusters_beside_background_gemms_hand_off_cleanly(cuda):
    """at H = 512 the four-wave clusters take 16 CUs each, so B = 128 bidirectional is 16 clusters = all 256 CUs
    of the chip, and the weight-gradient GEMMs of the layer above run beside every BPTT kernel on 128 workgroups that
    share those CUs.  Every cluster member must still become resident and no hand-off may time out (sticky error word 0
    after several training steps of a 3-layer model; the loss stays finite and falls)."""
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    rng = np.random.RandomState(12)
    B, T, D, C = 128, 48, 24, 12
    x = rng.randn(B, T, D).astype(np.float32)
    sl = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    sl[0] = T
    labels = np.full((B, 5), -1, dtype=np.int64)
    for b in range(B):
        x[b, sl[b]:] = 0
        labels[b, :4] = rng.randint(0, C, size=4)
    model = CTC('blstm', D, 512, 3, C, parameter_init=0.1, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=3)
    assert ops.check_async_errors(0) == 0
    losses = []
    for _ in range(6):
        loss, _ = model.compute_loss(x, labels, sl, keep_prob=0.9)
        model.train(loss, 'adam', 2e-3)
        losses.append(float(loss.item()))
    assert ops.check_async_errors(0) == 0
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses


def _grad_digest(model_fn, x, dense, sl, keep_prob=0.8):
    """loss + every gradient of one training step of a freshly built (seeded) model, as raw arrays."""
    model = model_fn()
    loss, _ = model.compute_loss(x, dense, sl, keep_prob=keep_prob)
    opt = model._set_optimizer('rmsprop', 1e-3)
    gv = opt.compute_gradients(loss, model=model)
    return float(loss.item()), {name: g.detach().cpu().numpy().copy() for g, name in gv}


def test_scheduling_switches_of_round5_do_not_change_a_bit(cuda, monkeypatch):
    """Where work is ISSUED must not show in the result: the LSTM layers' weight-gradient lanes behind the dx product
    (rnn_util.DW_AFTER_DX_FLOPS: never / always), the VGG front-end's weight-gradient kernels on a side lane
    (vgg_blstm.VGG_WGRAD_SIDE) and its forward in 2 / 3 runs of images on separate lanes (VGG_FWD_CHUNKS; dropout on, so
    the shifted dropout counters are covered) give the same loss and the same gradient bits as the forms they replace."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.models.encoders.core import rnn_util, vgg_blstm
    rng = np.random.RandomState(17)
    B, T, F, splice, C = 18, 30, 40, 11, 12
    x, sl, labs, dense = _batch(rng, B, T, F * 3 * splice, C)

    def build():
        return CTC('vgg_blstm', F * 3, 64, 3, C, splice=splice, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=2,
                   device='cuda:0')
    runs = {}
    monkeypatch.setattr(vgg_blstm, 'VGG_FWD_CHUNK_MIN', 1)       # the runs of images of the pipelined forward at this size
    for tag, flops, side, chunks in (('old', float('inf'), False, 1), ('after_dx', 0.0, False, 1),
                                     ('side', float('inf'), True, 1), ('both', 0.0, True, 1),
                                     ('two_runs', float('inf'), False, 2), ('all_three_runs', 0.0, True, 3)):
        monkeypatch.setattr(rnn_util, 'DW_AFTER_DX_FLOPS', flops)
        monkeypatch.setattr(vgg_blstm, 'VGG_WGRAD_SIDE', side)
        monkeypatch.setattr(vgg_blstm, 'VGG_FWD_CHUNKS', chunks)
        runs[tag] = _grad_digest(build, x, dense, sl)
    l0, g0 = runs['old']
    for tag in ('after_dx', 'side', 'both', 'two_runs', 'all_three_runs'):
        l, g = runs[tag]
        assert l == l0, (tag, l, l0)
        for n in g0:
            assert np.array_equal(g[n], g0[n]), (tag, n)


def test_attention_backward_products_on_bf16_operands_stay_within_bf16_rounding(cuda, monkeypatch):
    """ASR_ATT_BWD_BF16 (default for a bf16-operand model): the batched products of the backward pass outside the decoder
    loop round their operands to bf16.  Same loss and logits bit for bit (the forward is untouched), every gradient within
    bf16 operand rounding of the fp32-product form (relative L2 per variable)."""
    from tensorflow_end2end_speech_recognition_amd.models.attention import attention_seq2seq as S
    from tensorflow_end2end_speech_recognition_amd.models.attention.joint_ctc_attention import JointCTCAttention
    rng = np.random.RandomState(23)
    B, T, D, C, To = 8, 60, 24, 20, 9
    sl = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    sl[0] = T
    x = rng.randn(B, T, D).astype(np.float32)
    lens = rng.randint(3, To - 1, size=B)
    labels = np.full((B, To), C + 1, dtype=np.int64)
    ctc = np.full((B, To - 2), -1, dtype=np.int64)
    for b in range(B):
        x[b, sl[b]:] = 0
        y = rng.randint(0, C, size=lens[b])
        labels[b, 0] = C
        labels[b, 1:1 + lens[b]] = y
        ctc[b, :lens[b]] = y
    out = {}
    for flag in (False, True):
        monkeypatch.setattr(S, 'ATT_BWD_BF16', flag)
        m = JointCTCAttention(input_size=D, encoder_type='blstm', encoder_num_units=64, encoder_num_layers=2,
                              encoder_num_proj=None, attention_type='location', attention_dim=32, decoder_type='lstm',
                              decoder_num_units=64, decoder_num_layers=1, embedding_dim=16, lambda_weight=0.5,
                              num_classes=C, sos_index=C, eos_index=C + 1, max_decode_length=To + 3, parameter_init=0.1,
                              clip_grad_norm=5.0, clip_activation_encoder=50, clip_activation_decoder=50, dtype='bf16',
                              seed=5, device='cuda:0')
        loss, logits, *_ = m.compute_loss(x, labels, ctc, sl, lens + 2, 1.0, 1.0, 1.0)
        opt = m._set_optimizer('adam', 1e-3)
        gv = opt.compute_gradients(loss, model=m)
        out[flag] = (float(loss.item()), logits.detach().cpu().numpy().copy(),
                     {name: g.detach().cpu().numpy().copy() for g, name in gv})
    assert out[False][0] == out[True][0]
    assert np.array_equal(out[False][1], out[True][1])
    worst = 0.0
    for n, g in out[False][2].items():
        den = np.linalg.norm(g)
        if den > 1e-8:
            worst = max(worst, float(np.linalg.norm(out[True][2][n] - g) / den))
    assert 0.0 < worst < 2e-2, worst

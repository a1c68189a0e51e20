"""The oracle against the REFERENCE'S OWN CODE.

tests/golden/tfshim_v1.npz holds what the unchanged files of the reference (attention_layer.py, attention_decoder.py,
dynamic_decoder.py, bridge.py, attention_seq2seq.py, joint_ctc_attention.py, ctc.py, the encoders, the Python LSTMCell,
model_base.py) computed when executed on an eager float64 TensorFlow stand-in (tests/golden/make_golden_tfshim.py +
tests/golden/tf_shim; run where /root/reference exists).  These tests hold oracle/attention.py, oracle/lstm.py,
oracle/model.py, oracle/gru.py, oracle/vgg.py and oracle/optim.py to those numbers in float64: values 1e-11, gradients
1e-9 relative to the largest entry.  CPU only; nothing here touches the HIP path -- the `-m gpu` suite compares THAT
with the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import attention as oatt
from oracle import lstm as olstm
from oracle import model as omodel
from oracle import optim as ooptim

import _tfshim

F64 = torch.float64
VAL, GRAD = 1e-11, 1e-9


def _t(a, grad=False):
    return torch.as_tensor(np.asarray(a), dtype=F64).clone().requires_grad_(grad)


def _close(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # relative to the largest entry, with a floor: a gradient that is zero in exact arithmetic (uniform weights under
    # location attention) is rounding noise of 1e-17 on both sides
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6) if a.size else 0.0
    assert err < tol, (what, err)
    return err


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', _tfshim.cases('attention_layer'))
def test_tfshim_attention_layer(name):
    """models/attention/decoders/attention_layer.py:45-347 executed; oracle.attention.attention_step must agree on the
    weights, the context and the gradient of a random functional w.r.t. encoder outputs, decoder output, previous
    weights and every variable -- per attention type, with sharpening 2 (the float32.min mask TIMES the factor), with
    sigmoid smoothing, with ragged lengths down to one frame."""
    c = _tfshim.Case(name)
    m, I, O = c.meta, c.group('in'), c.group('out')
    pre = 'attention_layer/'
    P = {k[len(pre):]: _t(v, True) for k, v in c.variables().items()}
    enc, dec, pa = _t(I['enc'], True), _t(I['dec'], True), _t(I['prev_alpha'], True)
    lens = torch.as_tensor(I['lens'], dtype=torch.long)
    at = m['attention_type']
    keys = oatt.compute_keys(P, at, enc)
    uses_prev = at in ('location', 'hybrid')
    alpha, ctx = oatt.attention_step(P, at, enc, keys, dec, lens, m['sharpening'], m['sigmoid_smoothing'],
                                     pa if uses_prev else None)
    _close(alpha.detach(), O['alpha'], VAL, 'alpha')
    _close(ctx.detach(), O['ctx'], VAL, 'ctx')
    assert np.all(O['alpha'][np.arange(3)[:, None], np.arange(13)[None]][np.arange(13)[None] >= I['lens'][:, None]] == 0)
    f = (alpha * _t(I['r_alpha'])).sum() + (ctx * _t(I['r_ctx'])).sum()
    f.backward()
    GI = c.group('grad_in')
    _close(enc.grad, GI['enc'], GRAD, 'd enc')
    _close(dec.grad if dec.grad is not None else torch.zeros_like(dec), GI['dec'], GRAD, 'd dec')
    if 'prev_alpha' in GI:
        _close(pa.grad, GI['prev_alpha'], GRAD, 'd prev_alpha')
    got = {pre + k: (v.grad.numpy() if v.grad is not None else None) for k, v in P.items()}
    for n in list(got):
        if got[n] is None:
            assert n in m['none_grads'], n                       # e.g. W_keys under location attention (Q6)
            del got[n]
    c.check_grads(got, GRAD)
    if uses_prev and not m['carried']:
        # the reference's effective graph (Q1): zeros in -> the location features are the W_filter bias
        a2, c2 = oatt.attention_step(P, at, enc, keys, dec, lens, m['sharpening'], m['sigmoid_smoothing'], None)
        _close(a2.detach(), O['alpha'], VAL, 'alpha (prev=None)')
        _close(c2.detach(), O['ctx'], VAL, 'ctx (prev=None)')


@pytest.mark.parametrize('name', _tfshim.cases('lstm_cell'))
def test_tfshim_python_lstm_cell(name):
    """models/recurrent/layers/lstm.py:104-170 executed (the reference's own statement of the peephole / clip /
    projection cell): oracle.lstm.lstm_block_cell(clip_blocks_gradient=True) and lstmp_cell must agree on c, m and on the
    gradients -- with a clamp that is active for a part of the states (tf.clip_by_value: no gradient through them)."""
    c = _tfshim.Case(name)
    m, I, O = c.meta, c.group('in'), c.group('out')
    V = {k: _t(v, True) for k, v in c.variables().items()}
    H = I['c_prev'].shape[1]
    x, c0, m0 = _t(I['x'], True), _t(I['c_prev'], True), _t(I['m_prev'], True)
    z = torch.zeros(H, dtype=F64)
    peep = m['use_peephole']
    wci, wcf, wco = (V['lstm_cell/w_i_diag'], V['lstm_cell/w_f_diag'], V['lstm_cell/w_o_diag']) if peep else (z, z, z)
    clip = m['cell_clip'] or 0.0
    if m['num_proj']:
        cs, out = olstm.lstmp_cell(x, c0, m0, V['lstm_cell/kernel'], V['lstm_cell/bias'], wci, wcf, wco,
                                   V['lstm_cell/projection/kernel'], 1.0, clip, peep)
    else:
        cs, out = olstm.lstm_block_cell(x, c0, m0, V['lstm_cell/kernel'], V['lstm_cell/bias'], wci, wcf, wco,
                                        1.0, clip, peep, clip_blocks_gradient=True)
    _close(cs.detach(), O['c'], VAL, 'c')
    _close(out.detach(), O['m'], VAL, 'm')
    ((cs * _t(I['r_c'])).sum() + (out * _t(I['r_m'])).sum()).backward()
    GI = c.group('grad_in')
    _close(x.grad, GI['x'], GRAD, 'dx')
    _close(c0.grad, GI['c_prev'], GRAD, 'dc_prev')
    _close(m0.grad, GI['m_prev'], GRAD, 'dm_prev')
    c.check_grads({k: v.grad.numpy() for k, v in V.items()}, GRAD)
    if clip:
        # the fused LSTMBlockCell path of the oracle (straight-through clamp) has the SAME forward
        cs2, h2 = olstm.lstm_block_cell(_t(I['x']), _t(I['c_prev']), _t(I['m_prev']), V['lstm_cell/kernel'].detach(),
                                        V['lstm_cell/bias'].detach(), wci.detach(), wcf.detach(), wco.detach(),
                                        1.0, clip, peep)
        _close(cs2, O['c'], VAL, 'c (straight-through form)')
        assert 0.1 < m['clamped_fraction'] < 0.9


def _ctc_rows(I, prefix='labels'):
    flat, lens = I[prefix + '_flat'], I[prefix + '_len']
    out, k = [], 0
    for n in lens:
        out.append([int(v) for v in flat[k:k + n]])
        k += n
    return out


@pytest.mark.parametrize('name', _tfshim.cases('seq2seq'))
def test_tfshim_attention_models(name):
    """AttentionSeq2Seq.compute_loss / JointCTCAttention.compute_loss executed end to end (attention_seq2seq.py:193-664,
    attention_decoder.py, dynamic_decoder.py, bridge.py:128-151, joint_ctc_attention.py:182-346, blstm.py:258-332):
    oracle.attention.attention_model_forward must give the same teacher-forced logits, ids, attention weights, loss and
    the same gradient for EVERY variable, attention_model_infer the same greedy ids -- in both readings of the
    previous-weights attribute (prev_alpha zeros = the traced graph, carry = the eager run)."""
    c = _tfshim.Case(name)
    m, I, O = c.meta, c.group('in'), c.group('out')
    sd = c.variables()
    joint = m['lambda_weight'] is not None
    q2 = name.endswith('_q2')
    kw = dict(clip_enc=m['clip_enc'], clip_dec=m['clip_dec'], sharpening=m['sharpening'], temperature=m['temperature'],
              sigmoid_smoothing=m['sigmoid_smoothing'], prev_alpha=m['prev_alpha'])
    if joint:
        kw.update(ctc_labels=_ctc_rows(I, 'ctc_labels'), lambda_weight=m['lambda_weight'])
    out = oatt.attention_model_forward(sd, I['inputs'], I['labels'], I['inputs_seq_len'], I['labels_seq_len'],
                                       m['enc_layers'], m['attention_type'], **kw)
    B, To = O['train_predicted_ids'].shape
    _close(out['logits'] * m['temperature'], O['train_logits'], VAL, 'train logits')
    _close(out['logits'] + 1e-10, O['logits_returned'], VAL, 'returned logits')
    _close(out['alphas'], O['train_attention_weights'], VAL, 'attention weights')
    assert np.array_equal(out['predicted_ids'], O['train_predicted_ids'])
    T = I['inputs'].shape[1]
    if joint:
        # quirk Q2: the reference reshapes the BATCH-major [B*T, C] head output as if it were time-major
        ref = O['ctc_logits'].reshape(B, T, -1).transpose(1, 0, 2)
        _close(out['ctc_logits'], ref, VAL, 'ctc head logits (modulo the Q2 reshape)')
    if not q2:
        assert abs(out['total_loss'] - float(O['total_loss'])) < VAL * max(1.0, abs(float(O['total_loss'])))
        got = dict(out['grads'])
        for n in m['none_grads']:
            assert np.abs(got[n]).max() == 0.0, n
        c.check_grads(got, GRAD)
    ids = oatt.attention_model_infer(sd, I['inputs'], I['inputs_seq_len'], m['enc_layers'], m['attention_type'],
                                     m['sos'], m['eos'], m['max_decode_length'], clip_enc=m['clip_enc'],
                                     clip_dec=m['clip_dec'], sharpening=m['sharpening'],
                                     sigmoid_smoothing=m['sigmoid_smoothing'], prev_alpha=m['prev_alpha'])
    ref_ids = O['infer_predicted_ids']
    n = min(ids.shape[1], ref_ids.shape[1])
    assert np.array_equal(ids[:, :n], ref_ids[:, :n]) and not ref_ids[:, n:].any() and not ids[:, n:].any()


@pytest.mark.parametrize('name', _tfshim.cases('ctc'))
def test_tfshim_ctc_models(name):
    """CTC.compute_loss executed (models/ctc/ctc.py:175-323 over blstm.py / lstm.py / gru.py / vgg_blstm.py + cnn_util.py;
    the CTC loss itself is torch's, a third implementation): oracle.model must agree on encoder outputs, logits, loss and
    every gradient; ModelBase._clip_gradients (model_base.py:148-166) on oracle.optim.clip_by_norm."""
    c = _tfshim.Case(name)
    m, I, O = c.meta, c.group('in'), c.group('out')
    sd = c.variables()
    rows = _ctc_rows(I)
    et = m['encoder_type']
    clip = m['clip_activation'] or 0.0
    if et in ('gru', 'bgru'):
        out = omodel.gru_ctc_model_forward(sd, I['inputs'], rows, I['inputs_seq_len'], m['num_layers'],
                                           ndir=2 if et == 'bgru' else 1)
    elif m['lstm_impl'] == 'LSTMCell':
        out = omodel.lstmp_ctc_model_forward(sd, I['inputs'], rows, I['inputs_seq_len'], m['num_layers'],
                                             cell_clip=clip)
    else:
        vgg = (m['input_size'] // 3, m['splice']) if et == 'vgg_blstm' else None
        out = omodel.ctc_model_forward(sd, I['inputs'], rows, I['inputs_seq_len'], m['num_layers'],
                                       ndir=1 if et == 'lstm' else 2, cell_clip=clip,
                                       weight_decay=m['weight_decay'], temperature=m['temperature'], vgg=vgg,
                                       bottleneck=m['bottleneck'])
    _close(out['enc'], O['encoder_outputs'], VAL, 'encoder outputs')
    _close(out['logits'], O['logits'], VAL, 'logits')
    assert abs(out['total_loss'] - float(O['total_loss'])) < VAL * max(1.0, abs(float(O['total_loss'])))
    c.check_grads(out['grads'], GRAD)
    clipped = {n: np.asarray(ooptim.clip_by_norm(np.asarray(g), m['clip_grad_norm'])) for n, g in out['grads'].items()}
    c.check_grads(clipped, GRAD, group='clipped')
    norms = [np.sqrt((g ** 2).sum()) for g in out['grads'].values()]
    assert any(v > m['clip_grad_norm'] for v in norms)          # the clip is active


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference tree is only present in the build container')
def test_tfshim_fixture_is_what_the_generator_produces(tmp_path):
    """Where /root/reference exists: re-run tests/golden/make_golden_tfshim.py (the reference's model code on the eager
    TensorFlow stand-in) and compare every array with the committed fixture -- the fixture is a pure function of the
    reference's files, the stand-in and the generator's seeds."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'regen.npz')
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'golden', 'make_golden_tfshim.py'), '--out', out],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    new, old = np.load(out), np.load(_tfshim.PATH)
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if k == 'meta_json':
            assert bytes(new[k]) == bytes(old[k])
        else:
            assert new[k].shape == old[k].shape and np.allclose(new[k], old[k], rtol=1e-13, atol=1e-300), k

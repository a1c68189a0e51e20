"""Runs the reference's OWN model code (unchanged files under /root/reference) on the eager TensorFlow stand-in of
tests/golden/tf_shim and records what it computes: tests/golden/tfshim_v1.npz.

    python tests/golden/make_golden_tfshim.py            # needs /root/reference; the GPU box never runs this

What is executed (reference file:line) and what the fixture pins (tests/test_oracle.py::test_tfshim_*):
  * models/attention/decoders/attention_layer.py:45-347 -- AttentionLayer.__call__ for the seven implemented attention
    types x {sharpening 1, 2} x {softmax, sigmoid smoothing}, ragged lengths, zero and non-zero previous weights:
    attention weights, context, and the gradients of a random linear functional w.r.t. every input and variable
    -> oracle.attention.attention_step / compute_keys / location_features
  * models/recurrent/layers/lstm.py:104-170 -- the reference's Python LSTMCell: peepholes, an ACTIVE cell clip
    (tf.clip_by_value), projection -> oracle.lstm.lstm_block_cell(clip_blocks_gradient=True), oracle.lstm.lstmp_cell;
    and the stand-in's LSTMBlockCell is checked against it here (same forward)
  * models/attention/attention_seq2seq.py:193-664 + attention_decoder.py + dynamic_decoder.py + bridge.py +
    models/encoders/core/blstm.py -- AttentionSeq2Seq.compute_loss end to end per attention type (teacher-forced logits,
    ids, attention weights, greedy inference ids, sequence loss, the gradient of every variable), in the eager
    ("carry") and the traced-once ("zeros") reading of `self.attention_weights`
    -> oracle.attention.attention_model_forward / attention_model_infer
  * models/attention/joint_ctc_attention.py:182-346 -- JointCTCAttention.compute_loss: (1 - lambda) * sequence loss +
    lambda * mean CTC; B = 1 (where the reference's [B*T, C] -> [T, B, C] reshape, quirk Q2, is harmless) for loss and
    gradients, B = 3 for the CTC head's logits modulo that reshape
  * models/ctc/ctc.py:175-323 + models/encoders/core/{blstm,lstm,vgg_blstm,cnn_util}.py -- CTC.compute_loss for blstm
    (peephole + clip, weight decay, temperature), lstm (MultiRNNCell), blstm + bottleneck, vgg_blstm, blstm with
    lstm_impl='LSTMCell' + num_proj (the reference's Python LSTMCell registered as tf.contrib.rnn.LSTMCell), bgru
    -> oracle.model.ctc_model_forward / lstmp_ctc_model_forward / gru_ctc_model_forward
  * models/model_base.py:148-166 -- ModelBase._clip_gradients over optimizer.compute_gradients -> oracle.optim.clip_by_norm

Variables are created by the reference's own scopes / initializers on the first pass (their NAMES are part of the
fixture: they are the checkpoint layout of SURVEY Appendix C), then overwritten with seeded random values (so biases
and peepholes are non-trivial) and the model is evaluated again; that second pass is what is recorded.
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'tf_shim'))
sys.path.insert(0, '/root/reference')

import tensorflow as tf                                    # noqa: E402  (the stand-in)
import torch                                               # noqa: E402

from models.attention.decoders.attention_layer import AttentionLayer          # noqa: E402
from models.attention.decoders import attention_decoder as ref_attention_decoder  # noqa: E402
from models.attention.attention_seq2seq import AttentionSeq2Seq               # noqa: E402
from models.attention.joint_ctc_attention import JointCTCAttention            # noqa: E402
from models.ctc.ctc import CTC                                                 # noqa: E402
from models.recurrent.layers.lstm import LSTMCell as RefLSTMCell              # noqa: E402

tf.contrib.rnn.LSTMCell = RefLSTMCell      # "taken directly from TensorFlow code" (models/recurrent/layers/lstm.py:4)

OUT = {}
META = {}
META_VARS = {}

# --- the traced-once reading of AttentionDecoder.step (Q1): put the attribute back before every step ------------
_MODE = {'prev_alpha': 'carry'}
_orig_step = ref_attention_decoder.AttentionDecoder.step


def _step(self, time, inputs, state, name=None):
    if _MODE['prev_alpha'] == 'zeros':
        self.attention_weights = tf.zeros_like(self.attention_weights)
    return _orig_step(self, time, inputs, state, name)


ref_attention_decoder.AttentionDecoder.step = _step


def T(a, kind=None):
    return tf.convert_to_tensor(np.asarray(a))


def put(case, group, name, value):
    OUT['%s|%s|%s' % (case, group, name)] = np.asarray(value)


BIG = 20000          # elements; larger tensors (the hard-coded 64 / 128-channel VGG filters) are stored subsampled
STRIDE = 97
VARS = {}            # variable name -> [shape, seed, scale] of the case being generated


def var_values(shape, seed, scale):
    """The recipe the tests repeat (tests/_tfshim.py): a variable is uniform(-scale, scale) from its own seed."""
    return np.random.RandomState(seed).uniform(-scale, scale, size=tuple(shape))


def randomize(case, scale=0.5, bias_scale=0.3, overrides=None):
    VARS.clear()
    for name, v in tf.shim_variables().items():
        if not v._t.dtype.is_floating_point:
            continue
        shape = [int(d) for d in v._t.shape]
        s = bias_scale if ('bias' in name.lower()) else scale
        if len(shape) == 4 or name == 'bridge/weights':            # convolutions / the VGG bridge: fan-in aware
            s = 1.5 / np.sqrt(np.prod(shape[:-1]))
        if overrides and name in overrides:
            s = overrides[name]
        seed = zlib.crc32((case + '|' + name).encode()) & 0x7fffffff
        tf.shim_set_variable(name, var_values(shape, seed, s))
        VARS[name] = [shape, seed, float(s)]


def put_maybe_big(case, group, name, a):
    a = np.asarray(a)
    if a.size > BIG:
        put(case, group + '_sub', name, a.reshape(-1)[::STRIDE])
        put(case, group + '_stat', name, np.array([a.sum(), (a * a).sum()]))
    else:
        put(case, group, name, a)


def dump_vars(case):
    META_VARS[case] = {k: list(v) for k, v in VARS.items()}


def dump_grads(case, loss):
    tf.shim_zero_grads()
    loss._t.backward()
    none = []
    for name, v in tf.shim_variables().items():
        if not v._t.dtype.is_floating_point:
            continue
        if v._t.grad is None:
            none.append(name)
        else:
            put_maybe_big(case, 'grad', name, v._t.grad.numpy())
    return none


def sparse(labels_list):
    idx, val = [], []
    for b, row in enumerate(labels_list):
        for j, v in enumerate(row):
            idx.append([b, j])
            val.append(v)
    L = max([len(r) for r in labels_list] + [1])
    return tf.SparseTensor(T(np.asarray(idx, dtype=np.int64).reshape(-1, 2)), T(np.asarray(val, dtype=np.int64)),
                           T(np.asarray([len(labels_list), L], dtype=np.int64)))


# ------------------------------------------------------------------------------------------------------------------
def attention_layer_cases():
    """attention_layer.py:45-347, one call of the layer."""
    types = ['bahdanau_content', 'location', 'hybrid', 'dot_product', 'luong_dot', 'luong_general', 'luong_concat']
    B, Tn, A = 3, 13, 5
    lens = np.array([13, 7, 1])
    k = 0
    for at in types:
        for sharp in (1.0, 2.0):
            for smooth in (False, True):
                for prev in ('zeros', 'random'):
                    if prev == 'random' and at not in ('location', 'hybrid'):
                        continue
                    k += 1
                    rng = np.random.RandomState(1000 + k)
                    E = 6
                    U = E if at == 'luong_dot' else 4
                    case = 'attlayer_%s_s%g_%s_%s' % (at, sharp, 'sig' if smooth else 'soft', prev)
                    enc = rng.randn(B, Tn, E)
                    dec = rng.randn(B, U)
                    if prev == 'zeros':
                        pa = np.zeros((B, Tn))
                    else:
                        pa = rng.rand(B, Tn) * (np.arange(Tn)[None] < lens[:, None])
                        pa /= pa.sum(1, keepdims=True)
                    r_alpha, r_ctx = rng.randn(B, Tn), rng.randn(B, E)

                    def run():
                        layer = AttentionLayer(at, A, 0.1, sharp, smooth, tf.contrib.learn.ModeKeys.TRAIN)
                        e, d, p = T(enc), T(dec), T(pa)
                        for t_ in (e, d, p):
                            t_._t.requires_grad_(True)
                        alpha, ctx = layer(e, d, T(lens), p)
                        return e, d, p, alpha, ctx

                    tf.shim_reset(seed=k)
                    run()
                    randomize(case, 0.8, 0.5)
                    tf.shim_reset(keep_variables=True)
                    e, d, p, alpha, ctx = run()
                    dump_vars(case)
                    f = tf.reduce_sum(alpha * T(r_alpha)) + tf.reduce_sum(ctx * T(r_ctx))
                    none = dump_grads(case, f)
                    for nm, val in (('enc', enc), ('dec', dec), ('prev_alpha', pa), ('lens', lens),
                                    ('r_alpha', r_alpha), ('r_ctx', r_ctx)):
                        put(case, 'in', nm, val)
                    put(case, 'out', 'alpha', alpha.numpy())
                    put(case, 'out', 'ctx', ctx.numpy())
                    put(case, 'grad_in', 'enc', e._t.grad.numpy())
                    put(case, 'grad_in', 'dec', d._t.grad.numpy() if d._t.grad is not None else np.zeros_like(dec))
                    if p._t.grad is not None:
                        put(case, 'grad_in', 'prev_alpha', p._t.grad.numpy())
                    META[case] = dict(kind='attention_layer', attention_type=at, sharpening=sharp,
                                      sigmoid_smoothing=smooth, num_units=A, none_grads=none,
                                      carried=prev == 'random')


def lstm_cell_cases():
    """models/recurrent/layers/lstm.py:104-170."""
    B, D, H = 4, 3, 5
    k = 0
    for peep in (False, True):
        for clip in (None, 0.3):
            for proj in (None, 3):
                k += 1
                rng = np.random.RandomState(2000 + k)
                case = 'lstmcell_p%d_c%s_r%s' % (peep, clip, proj)
                P = proj or H
                x, c0, m0 = rng.randn(B, D), rng.randn(B, H) * 0.6, rng.randn(B, P) * 0.6
                r_c, r_m = rng.randn(B, H), rng.randn(B, P)

                def run():
                    cell = RefLSTMCell(H, use_peepholes=peep, cell_clip=clip, num_proj=proj, forget_bias=1.0)
                    xs = [T(x), T(c0), T(m0)]
                    for t_ in xs:
                        t_._t.requires_grad_(True)
                    out, (c, m) = cell(xs[0], tf.contrib.rnn.LSTMStateTuple(xs[1], xs[2]))
                    return xs, out, c, m

                tf.shim_reset(seed=k)
                run()
                randomize(case, 0.9, 0.5)
                tf.shim_reset(keep_variables=True)
                xs, out, c, m = run()
                assert out is m
                if clip is not None:
                    frac = float((np.abs(c.numpy()) >= clip - 1e-12).mean())
                    assert 0.1 < frac < 0.9, frac           # the clamp must be ACTIVE and not everywhere
                else:
                    frac = 0.0
                dump_vars(case)
                f = tf.reduce_sum(c * T(r_c)) + tf.reduce_sum(m * T(r_m))
                dump_grads(case, f)
                for nm, val in (('x', x), ('c_prev', c0), ('m_prev', m0), ('r_c', r_c), ('r_m', r_m)):
                    put(case, 'in', nm, val)
                put(case, 'out', 'c', c.numpy())
                put(case, 'out', 'm', m.numpy())
                for nm, t_ in zip(('x', 'c_prev', 'm_prev'), xs):
                    put(case, 'grad_in', nm, t_._t.grad.numpy())
                META[case] = dict(kind='lstm_cell', use_peephole=peep, cell_clip=clip, num_proj=proj,
                                  clamped_fraction=frac)
                if proj is None:
                    # the stand-in's LSTMBlockCell against the reference's statement of the same cell (forward)
                    blk = tf.contrib.rnn.LSTMBlockCell(H, forget_bias=1.0, clip_cell=clip, use_peephole=peep)
                    tf.shim_reset(keep_variables=True)
                    h2, (c2, _) = blk(T(x), tf.contrib.rnn.LSTMStateTuple(T(c0), T(m0)))
                    assert np.abs(h2.numpy() - m.numpy()).max() < 1e-14 and np.abs(c2.numpy() - c.numpy()).max() < 1e-14


def make_attention_batch(rng, B, Tn, D, C, Lmax):
    lens = np.sort(rng.randint(max(2, Tn // 2), Tn + 1, size=B))[::-1].copy()
    lens[0] = Tn
    x = rng.randn(B, Tn, D) * (np.arange(Tn)[None, :, None] < lens[:, None, None])
    sos, eos = C, C + 1
    lab_len = rng.randint(1, Lmax + 1, size=B)
    lab_len[rng.randint(B)] = Lmax
    labels = np.full((B, Lmax + 2), eos, dtype=np.int64)
    ctc_rows = []
    for b in range(B):
        y = rng.randint(0, C, size=lab_len[b])
        labels[b, 0] = sos
        labels[b, 1:1 + lab_len[b]] = y
        ctc_rows.append([int(v) for v in y])
    return x, lens, labels, lab_len + 2, ctc_rows, sos, eos


def seq2seq_cases():
    """AttentionSeq2Seq.compute_loss / JointCTCAttention.compute_loss end to end."""
    cfgs = []
    for at in ['bahdanau_content', 'location', 'hybrid', 'dot_product', 'luong_dot', 'luong_general', 'luong_concat']:
        modes = ('zeros', 'carry') if at in ('location', 'hybrid') else ('zeros',)
        for pm in modes:
            cfgs.append(dict(name='seq2seq_%s_%s' % (at, pm), at=at, pm=pm, sharp=1.0, temp=1.0, smooth=False,
                             joint=None, B=3))
    cfgs.append(dict(name='seq2seq_bahdanau_content_sharp2_temp2', at='bahdanau_content', pm='zeros', sharp=2.0,
                     temp=2.0, smooth=False, joint=None, B=3))
    cfgs.append(dict(name='seq2seq_hybrid_carry_sigmoid', at='hybrid', pm='carry', sharp=1.0, temp=1.0, smooth=True,
                     joint=None, B=3))
    cfgs.append(dict(name='joint_location_zeros_B1', at='location', pm='zeros', sharp=1.0, temp=1.0, smooth=False,
                     joint=0.5, B=1))
    cfgs.append(dict(name='joint_hybrid_carry_B1', at='hybrid', pm='carry', sharp=2.0, temp=2.0, smooth=False,
                     joint=0.3, B=1))
    cfgs.append(dict(name='joint_location_zeros_B3_q2', at='location', pm='zeros', sharp=1.0, temp=1.0, smooth=False,
                     joint=0.5, B=3))
    for k, cf in enumerate(cfgs):
        rng = np.random.RandomState(3000 + k)
        B, Tn, D, C, Lmax = cf['B'], 11, 6, 5, 4
        H, layers_n, A, E = 4, 2, 5, 3
        U = 2 * H if cf['at'] == 'luong_dot' else 6
        x, lens, labels, lab_lens, ctc_rows, sos, eos = make_attention_batch(rng, B, Tn, D, C, Lmax)
        max_dec = 7
        kw = dict(input_size=D, encoder_type='blstm', encoder_num_units=H, encoder_num_layers=layers_n,
                  encoder_num_proj=None, attention_type=cf['at'], attention_dim=A, decoder_type='lstm',
                  decoder_num_units=U, decoder_num_layers=1, embedding_dim=E, num_classes=C, sos_index=sos,
                  eos_index=eos, max_decode_length=max_dec, lstm_impl='LSTMBlockCell', use_peephole=True,
                  parameter_init=0.1, clip_grad_norm=5.0, clip_activation_encoder=0.6, clip_activation_decoder=0.6,
                  sharpening_factor=cf['sharp'], logits_temperature=cf['temp'])
        _MODE['prev_alpha'] = cf['pm']

        def run():
            if cf['joint'] is None:
                model = AttentionSeq2Seq(sigmoid_smoothing=cf['smooth'], **kw)
                total, logits, d_train, d_infer = model.compute_loss(
                    T(x), T(labels), T(lens), T(lab_lens), 1.0, 1.0, 1.0)
                return model, total, logits, None, d_train, d_infer
            model = JointCTCAttention(lambda_weight=cf['joint'], **kw)
            total, logits, ctc_logits, d_train, d_infer = model.compute_loss(
                T(x), T(labels), sparse(ctc_rows), T(lens), T(lab_lens), 1.0, 1.0, 1.0)
            return model, total, logits, ctc_logits, d_train, d_infer

        tf.shim_reset(seed=100 + k)
        run()
        case = cf['name']
        randomize(case, 0.7, 0.4)
        tf.shim_reset(keep_variables=True)
        model, total, logits, ctc_logits, d_train, d_infer = run()
        dump_vars(case)
        none = dump_grads(case, total)
        for nm, val in (('inputs', x), ('inputs_seq_len', lens), ('labels', labels), ('labels_seq_len', lab_lens)):
            put(case, 'in', nm, val)
        put(case, 'out', 'total_loss', total.numpy())
        put(case, 'out', 'logits_returned', logits.numpy())
        put(case, 'out', 'train_logits', d_train.logits.numpy())
        put(case, 'out', 'train_predicted_ids', d_train.predicted_ids.numpy())
        put(case, 'out', 'train_attention_weights', d_train.attention_weights.numpy())
        put(case, 'out', 'train_context_vector', d_train.context_vector.numpy())
        put(case, 'out', 'train_decoder_output', d_train.decoder_output.numpy())
        put(case, 'out', 'infer_predicted_ids', d_infer.predicted_ids.numpy())
        put(case, 'out', 'infer_attention_weights', d_infer.attention_weights.numpy())
        if ctc_logits is not None:
            put(case, 'out', 'ctc_logits', ctc_logits.numpy())
            put(case, 'in', 'ctc_labels_flat', np.asarray([v for r in ctc_rows for v in r], dtype=np.int64))
            put(case, 'in', 'ctc_labels_len', np.asarray([len(r) for r in ctc_rows], dtype=np.int64))
        # Q15: JointCTCAttention.__init__ hands clip_activation_decoder=50, weight_decay=0.0, time_major=True,
        # sharpening_factor=1.0, logits_temperature=1.0 to its base class whatever the caller passed
        # (joint_ctc_attention.py:133-137); the meta block records the EFFECTIVE values
        eff = cf['joint'] is not None
        META[case] = dict(kind='seq2seq', attention_type=cf['at'], prev_alpha=cf['pm'],
                          sharpening=1.0 if eff else cf['sharp'], temperature=1.0 if eff else cf['temp'],
                          sigmoid_smoothing=cf['smooth'], lambda_weight=cf['joint'],
                          ctor_args=dict(clip_activation_decoder=0.6, sharpening_factor=cf['sharp'],
                                         logits_temperature=cf['temp']),
                          enc_layers=layers_n, clip_enc=0.6, clip_dec=50.0 if eff else 0.6, sos=sos, eos=eos,
                          max_decode_length=max_dec, none_grads=none, num_classes=C)
    _MODE['prev_alpha'] = 'carry'


def ctc_cases():
    """CTC.compute_loss (models/ctc/ctc.py:175-323) over the encoders of the hot path."""
    cfgs = [
        dict(name='ctc_blstm', enc='blstm', kw=dict(lstm_impl='LSTMBlockCell', use_peephole=True,
                                                    clip_activation=0.5, num_proj=0, weight_decay=1e-3), temp=2),
        dict(name='ctc_blstm_nopeep_bottleneck', enc='blstm',
             kw=dict(lstm_impl='LSTMBlockCell', use_peephole=False, clip_activation=0.5, num_proj=0,
                     bottleneck_dim=4), temp=1),
        dict(name='ctc_lstm', enc='lstm', kw=dict(lstm_impl='LSTMBlockCell', use_peephole=True, clip_activation=0.5,
                                                  num_proj=0), temp=1),
        dict(name='ctc_blstm_lstmcell_proj', enc='blstm',
             kw=dict(lstm_impl='LSTMCell', use_peephole=True, clip_activation=0.5, num_proj=3), temp=1),
        dict(name='ctc_vgg_blstm', enc='vgg_blstm', kw=dict(lstm_impl='LSTMBlockCell', use_peephole=True,
                                                            clip_activation=0.5, num_proj=0, splice=3), temp=1),
        dict(name='ctc_bgru', enc='bgru', kw=dict(num_proj=0), temp=1),
        dict(name='ctc_gru', enc='gru', kw=dict(num_proj=0), temp=1),
    ]
    for k, cf in enumerate(cfgs):
        rng = np.random.RandomState(4000 + k)
        B, Tn, C, H, L = 4, 9, 5, 4, 2
        F = 12 if cf['enc'] == 'vgg_blstm' else 6
        D = F * cf['kw'].get('splice', 1)
        lens = np.array([9, 7, 4, 2])
        x = rng.randn(B, Tn, D) * (np.arange(Tn)[None, :, None] < lens[:, None, None])
        rows = [[1, 1, 3], [0, 4], [2], []]          # a repeat, an empty row
        kw = dict(encoder_type=cf['enc'], input_size=F, num_units=H, num_layers=L, num_classes=C,
                  parameter_init=0.1, clip_grad_norm=0.05)
        kw.update(cf['kw'])

        def run():
            model = CTC(**kw)
            total, logits = model.compute_loss(T(x), sparse(rows), T(lens), 1.0, softmax_temperature=cf['temp'])
            return model, total, logits

        tf.shim_reset(seed=200 + k)
        run()
        case = cf['name']
        randomize(case, 0.7, 0.4)
        tf.shim_reset(keep_variables=True)
        model, total, logits = run()
        dump_vars(case)
        # ModelBase._clip_gradients (model_base.py:148-166) on the gradients of this loss
        model.optimizer = tf.train.GradientDescentOptimizer(0.1)
        gv = model.optimizer.compute_gradients(total)
        clipped = model._clip_gradients(gv)
        for g, v in clipped:
            put_maybe_big(case, 'clipped', v.name[:-2], g.numpy())
        none = dump_grads(case, total)
        put(case, 'in', 'inputs', x)
        put(case, 'in', 'inputs_seq_len', lens)
        put(case, 'in', 'labels_flat', np.asarray([v for r in rows for v in r], dtype=np.int64))
        put(case, 'in', 'labels_len', np.asarray([len(r) for r in rows], dtype=np.int64))
        put(case, 'out', 'total_loss', total.numpy())
        put(case, 'out', 'logits', logits.numpy())
        put(case, 'out', 'encoder_outputs', model.encoder_outputs.numpy())
        META[case] = dict(kind='ctc', encoder_type=cf['enc'], num_layers=L, temperature=cf['temp'],
                          clip_activation=cf['kw'].get('clip_activation'), weight_decay=cf['kw'].get('weight_decay', 0.0),
                          bottleneck=bool(cf['kw'].get('bottleneck_dim')), splice=cf['kw'].get('splice', 1),
                          input_size=F, clip_grad_norm=0.05, none_grads=none,
                          lstm_impl=cf['kw'].get('lstm_impl'), num_proj=cf['kw'].get('num_proj', 0))


def main():
    attention_layer_cases()
    lstm_cell_cases()
    seq2seq_cases()
    ctc_cases()
    for c in META:
        META[c]['vars'] = META_VARS[c]
    OUT['meta_json'] = np.frombuffer(json.dumps(META, sort_keys=True).encode(), dtype=np.uint8)
    path = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else os.path.join(HERE, 'tfshim_v1.npz')
    np.savez_compressed(path, **OUT)
    print('%d arrays, %d cases -> %s (%.1f KB)' % (len(OUT), len(META), path, os.path.getsize(path) / 1024))
    for c in sorted(META):
        print(' ', c, META[c].get('none_grads') or '')


if __name__ == '__main__':
    main()

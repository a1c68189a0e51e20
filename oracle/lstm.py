"""Oracle (test infrastructure; the peephole-free cell is PINNED to TensorFlow's lstm_ops_test.py constants, the
peephole / clip / sequence-masking parts are PARITY UNPINNED -- see oracle/__init__.py):
CPU restatement of the recurrent encoder of the reference.

Follows
  * models/encoders/core/blstm.py:258-332  (lstmblockcell: per layer two
    LSTMBlockCell(forget_bias=1.0, clip_cell, use_peephole) wrapped in
    DropoutWrapper(output_keep_prob) -> bidirectional_dynamic_rnn -> concat)
  * models/encoders/core/lstm.py:241-304   (unidirectional MultiRNNCell form)
  * models/recurrent/layers/lstm.py:142-170 (the only in-repo statement of the
    peephole LSTM equations)
and, for what lives in TensorFlow itself, SURVEY.md Appendix B:
  * LSTMBlockCell: icfo = [x,h]W + b, column blocks i, ci, f, o;
    i = sig(i + wci*cs_prev); ci = tanh(ci); f = sig(f + fb + wcf*cs_prev);
    cs = ci*i + cs_prev*f; clip; o = sig(o + wco*cs); h = tanh(cs)*o.
    The TF gradient kernel (LSTMBlockCellGrad, attr use_peephole only) does
    not know about cell_clip, i.e. the clip is straight-through in backward.
  * dynamic_rnn(sequence_length): for t >= len[b] the output is zero and the
    state is copied through; the backward direction runs over the
    reverse_sequence'd valid prefix.
  * DropoutWrapper(output_keep_prob): mask/keep on the emitted output only,
    the recurrent (c, h) are untouched.

Written with torch (CPU) tensors so that gradients come from autograd; dtype is
whatever the caller passes (float64 for the checker).
"""
import torch


def lstm_block_cell(x, cs_prev, h_prev, w, b, wci, wcf, wco,
                    forget_bias=1.0, cell_clip=0.0, use_peephole=True):
    """One LSTMBlockCell step. x [B,Din], cs_prev/h_prev [B,H], w [Din+H,4H]."""
    H = cs_prev.shape[1]
    icfo = torch.cat([x, h_prev], dim=1) @ w + b
    i, ci, f, o = icfo[:, :H], icfo[:, H:2 * H], icfo[:, 2 * H:3 * H], icfo[:, 3 * H:]
    if use_peephole:
        i = i + wci * cs_prev
        f = f + wcf * cs_prev
    i = torch.sigmoid(i)
    ci = torch.tanh(ci)
    f = torch.sigmoid(f + forget_bias)
    cs = ci * i + cs_prev * f
    if cell_clip is not None and cell_clip > 0:
        # straight-through in backward (TF LSTMBlockCellGrad ignores the clip)
        cs = cs + (torch.clamp(cs, -cell_clip, cell_clip) - cs).detach()
    if use_peephole:
        o = o + wco * cs
    o = torch.sigmoid(o)
    h = torch.tanh(cs) * o
    return cs, h


def reverse_sequence(x_tm, seq_len):
    """tf.reverse_sequence(seq_axis=0, batch_axis=1): reverse the first len[b]
    frames of utterance b, leave the rest in place."""
    T, B = x_tm.shape[0], x_tm.shape[1]
    t = torch.arange(T).unsqueeze(1)                     # [T,1]
    L = seq_len.unsqueeze(0)                             # [1,B]
    src = torch.where(t < L, L - 1 - t, t)               # [T,B]
    return x_tm[src, torch.arange(B).unsqueeze(0)]


def dynamic_rnn(x_tm, seq_len, p, reverse=False, drop_mask=None,
                forget_bias=1.0, cell_clip=0.0, use_peephole=True):
    """tf.nn.dynamic_rnn(time_major=True, sequence_length) over one direction.

    x_tm [T,B,Din]; seq_len LongTensor [B]; p = dict(w,b,wci,wcf,wco).
    reverse=True gives the backward half of bidirectional_dynamic_rnn
    (reverse_sequence on the valid prefix, outputs reversed back).
    drop_mask [T,B,H] (already scaled by 1/keep_prob, indexed by FRAME) multiplies
    the emitted output only.  Returns out [T,B,H], (c_final, h_final).
    """
    T, B, _ = x_tm.shape
    H = p['b'].shape[0] // 4
    if reverse:
        x_tm = reverse_sequence(x_tm, seq_len)
    c = x_tm.new_zeros(B, H)
    h = x_tm.new_zeros(B, H)
    outs = []
    for s in range(T):
        active = (s < seq_len).to(x_tm.dtype).unsqueeze(1)
        c_new, h_new = lstm_block_cell(x_tm[s], c, h, p['w'], p['b'], p['wci'], p['wcf'],
                                       p['wco'], forget_bias, cell_clip, use_peephole)
        c = active * c_new + (1 - active) * c
        h = active * h_new + (1 - active) * h
        outs.append(h_new * active)
    out = torch.stack(outs, dim=0)
    if reverse:
        out = reverse_sequence(out, seq_len)
    if drop_mask is not None:
        out = out * drop_mask
    return out, (c, h)


def blstm_layer(x_tm, seq_len, p_fw, p_bw, drop_fw=None, drop_bw=None, **kw):
    """One 'blstm_hidden{i}' scope of blstm.py:281-323."""
    o_fw, st_fw = dynamic_rnn(x_tm, seq_len, p_fw, False, drop_fw, **kw)
    o_bw, st_bw = dynamic_rnn(x_tm, seq_len, p_bw, True, drop_bw, **kw)
    return torch.cat([o_fw, o_bw], dim=2), (st_fw, st_bw)


def blstm_encoder(inputs_bm, seq_len, layers, drop_masks=None, **kw):
    """BLSTMEncoder.__call__ (blstm.py:62-121) with time_major=True.

    inputs_bm [B,T,D] -> outputs [T,B,2H], final_state of the LAST layer.
    layers = [(p_fw, p_bw), ...]; drop_masks = [(m_fw, m_bw), ...] or None.
    """
    x = inputs_bm.transpose(0, 1)  # blstm.py:277-279
    final = None
    for li, (p_fw, p_bw) in enumerate(layers):
        m = drop_masks[li] if drop_masks is not None else (None, None)
        x, final = blstm_layer(x, seq_len, p_fw, p_bw, m[0], m[1], **kw)
    return x, final


def lstm_encoder(inputs_bm, seq_len, layers, drop_masks=None, **kw):
    """LSTMEncoder (lstm.py:241-304): MultiRNNCell of L cells in one dynamic_rnn.
    Mathematically identical to running the layers one after another because
    each layer's masking is the same sequence_length rule."""
    x = inputs_bm.transpose(0, 1)
    finals = []
    for li, p in enumerate(layers):
        m = drop_masks[li] if drop_masks is not None else None
        x, st = dynamic_rnn(x, seq_len, p, False, m, **kw)
        finals.append(st)
    return x, finals


def init_lstm_params(rng, din, H, init=0.1, dtype=torch.float64):
    """uniform(+-init) kernel & peepholes, zero bias (blstm.py:283-284, App. B)."""
    import numpy as np
    u = lambda *s: torch.tensor(rng.uniform(-init, init, size=s), dtype=dtype)
    return dict(w=u(din + H, 4 * H), b=torch.zeros(4 * H, dtype=dtype),
                wci=u(H), wcf=u(H), wco=u(H))

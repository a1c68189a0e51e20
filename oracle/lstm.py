"""Oracle (test infrastructure; PINNED -- the peephole-free cell to TensorFlow's lstm_ops_test.py constants; peepholes,
cell clip (both gradient conventions) and the projection to the reference's own Python LSTMCell
(models/recurrent/layers/lstm.py:104-170) as EXECUTED, and the stacked / bidirectional / masked encoders to the
reference's own encoder + model code as executed (tests/golden/tfshim_v1.npz, tests/test_oracle_tfshim.py; inside
those runs dynamic_rnn's zero-output / state-copy rule is the TensorFlow stand-in's statement of rnn.py) -- see
oracle/__init__.py):
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
                    forget_bias=1.0, cell_clip=0.0, use_peephole=True, clip_blocks_gradient=False, mm=None):
    """One LSTMBlockCell step. x [B,Din], cs_prev/h_prev [B,H], w [Din+H,4H].
    clip_blocks_gradient: the python LSTMCell (tf.contrib.rnn.LSTMCell, the projected cells of lstm_impl='LSTMCell',
    models/encoders/core/blstm.py:215-230 / models/recurrent/layers/lstm.py:152-157) clamps with tf.clip_by_value, whose
    gradient is zero where the state was clamped; the fused LSTMBlockCell's gradient op ignores the clip."""
    H = cs_prev.shape[1]
    xh = torch.cat([x, h_prev], dim=1)
    icfo = (xh @ w if mm is None else mm(xh, w)) + b           # mm: a product with its own backward (oracle.attention)
    i, ci, f, o = icfo[:, :H], icfo[:, H:2 * H], icfo[:, 2 * H:3 * H], icfo[:, 3 * H:]
    if use_peephole:
        i = i + wci * cs_prev
        f = f + wcf * cs_prev
    i = torch.sigmoid(i)
    ci = torch.tanh(ci)
    f = torch.sigmoid(f + forget_bias)
    cs = ci * i + cs_prev * f
    if cell_clip is not None and cell_clip > 0:
        if clip_blocks_gradient:
            cs = torch.clamp(cs, -cell_clip, cell_clip)
        else:
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


def ste_round(x, fn):
    """Straight-through rounding of a torch tensor: value fn(x), gradient of the identity."""
    return x + (fn(x.detach()) - x.detach())


def bf16_round_t(x):
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


def dynamic_rnn(x_tm, seq_len, p, reverse=False, drop_mask=None,
                forget_bias=1.0, cell_clip=0.0, use_peephole=True, h_round=None, clip_blocks_gradient=False):
    """tf.nn.dynamic_rnn(time_major=True, sequence_length) over one direction.
    clip_blocks_gradient: see lstm_block_cell (the python LSTMCell's tf.clip_by_value).

    x_tm [T,B,Din]; seq_len LongTensor [B]; p = dict(w,b,wci,wcf,wco).
    reverse=True gives the backward half of bidirectional_dynamic_rnn
    (reverse_sequence on the valid prefix, outputs reversed back).
    drop_mask [T,B,H] (already scaled by 1/keep_prob, indexed by FRAME) multiplies
    the emitted output only.  Returns out [T,B,H], (c_final, h_final).
    h_round (tests of the bf16-operand device path): rounding applied, straight-through, to the h
    that is fed back through W_h and emitted (the device keeps c and the final h unrounded).
    """
    T, B, _ = x_tm.shape
    H = p['b'].shape[0] // 4
    if reverse:
        x_tm = reverse_sequence(x_tm, seq_len)
    c = x_tm.new_zeros(B, H)
    h = x_tm.new_zeros(B, H)
    h_fb = h
    outs = []
    for s in range(T):
        active = (s < seq_len).to(x_tm.dtype).unsqueeze(1)
        c_new, h_new = lstm_block_cell(x_tm[s], c, h_fb, p['w'], p['b'], p['wci'], p['wcf'],
                                       p['wco'], forget_bias, cell_clip, use_peephole,
                                       clip_blocks_gradient=clip_blocks_gradient)
        c = active * c_new + (1 - active) * c
        h = active * h_new + (1 - active) * h
        h_emit = h_new if h_round is None else ste_round(h_new, h_round)
        h_fb = h if h_round is None else active * h_emit + (1 - active) * h_fb
        outs.append(h_emit * active)
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


# --------------------------------------------------------------------------------------------------
# Second, independent restatement (numpy, explicit BPTT -- no autograd) of ONE direction of a layer,
# with optional rounding of the MFMA operands.  Two uses (tests only):
#   * round_fn=None: fp64 hand-derived forward + backward of the same LSTMBlockCell / dynamic_rnn
#     semantics as above (peepholes, forget bias, straight-through cell clip, sequence_length masking,
#     reverse_sequence) -- checked against the autograd path in tests/test_oracle.py, so the two
#     statements of the peephole / clip / masking rules pin each other;
#   * round_fn=bf16_round: the rounding points of the bf16-operand device kernels (csrc/lstm_cluster.hip,
#     csrc/lstm.hip) are reproduced -- x, W_x, W_h, the fed-back / emitted h, the saved gate
#     activations and the gate gradients entering dG.W_h^T are bf16, everything else fp64 -- so the
#     device results can be held to ~1e-3 instead of "within bf16 noise of the fp64 oracle".
def bf16_round(a):
    """float array -> nearest-even bfloat16 -> float64 (numpy in, numpy out)."""
    import numpy as np
    t = torch.as_tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float64)
    return t.numpy()


def _sig(x):
    import numpy as np
    return 1.0 / (1.0 + np.exp(-x))


def layer_forward_np(x_tm, lens, p, reverse=False, forget_bias=1.0, cell_clip=0.0, use_peephole=True,
                     round_fn=None):
    """x_tm [T,B,Din] float64 numpy (already rounded by the caller if round_fn is used);
    p: dict of numpy arrays w [Din+H,4H] (columns i|ci|f|o), b, wci, wcf, wco.
    Returns dict: gates [T,B,4,H] (post-activation i, ci, f, o stored at the FRAME they belong to;
    rounded if round_fn), cs [T,B,H], hout [T,B,H] (zero past len; rounded), c_final, h_final
    (h_final unrounded fp32-like: the device keeps the final h in fp32)."""
    import numpy as np
    rnd = round_fn if round_fn is not None else (lambda a: a)
    T, B, Din = x_tm.shape
    H = p['b'].shape[0] // 4
    w = rnd(p['w'])
    wx, wh = w[:Din], w[Din:]
    wci, wcf, wco = (p['wci'], p['wcf'], p['wco']) if use_peephole else (0.0, 0.0, 0.0)
    lens = np.asarray(lens)
    xproj = x_tm.reshape(T * B, Din) @ wx + p['b']
    xproj = xproj.reshape(T, B, 4 * H)
    gates = np.zeros((T, B, 4, H))
    cs = np.zeros((T, B, H))
    hout = np.zeros((T, B, H))
    c = np.zeros((B, H))
    h = np.zeros((B, H))          # unrounded running h (the device's fp32 register copy)
    hb = np.zeros((B, H))         # what is fed back through W_h (rounded)
    rows = np.arange(B)
    for s in range(int(lens.max()) if B else 0):
        act = s < lens
        fr = np.where(act, (lens - 1 - s) if reverse else s, 0)
        pre = xproj[fr, rows] + hb @ wh
        i = _sig(pre[:, :H] + wci * c)
        g = np.tanh(pre[:, H:2 * H])
        f = _sig(pre[:, 2 * H:3 * H] + forget_bias + wcf * c)
        cn = g * i + c * f
        if cell_clip and cell_clip > 0:
            cn = np.clip(cn, -cell_clip, cell_clip)
        o = _sig(pre[:, 3 * H:] + wco * cn)
        hn = np.tanh(cn) * o
        a = act[:, None]
        c = np.where(a, cn, c)
        h = np.where(a, hn, h)
        hb = np.where(a, rnd(hn), hb)
        ra, fa = rows[act], fr[act]
        gates[fa, ra] = rnd(np.stack([i, g, f, o], 1))[act]
        cs[fa, ra] = cn[act]
        hout[fa, ra] = rnd(hn)[act]
    return dict(gates=gates, cs=cs, hout=hout, c_final=c, h_final=h, xproj=xproj)


def layer_backward_np(dout, gates, cs, lens, p, reverse=False, use_peephole=True, d_c_final=None,
                      d_h_final=None, round_fn=None):
    """Explicit BPTT of layer_forward_np (LSTMBlockCellGrad semantics: cell clip straight-through).
    dout [T,B,H] gradient w.r.t. the emitted outputs; gates/cs as saved by the forward (by frame).
    Returns dict: dgates [T,B,4,H] (pre-activation gradients i, ci, f, o by frame, zero past len; rounded
    if round_fn), dpeep [3,H] (wci, wcf, wco), db [4H]."""
    import numpy as np
    rnd = round_fn if round_fn is not None else (lambda a: a)
    T, B, H = dout.shape
    Din = p['w'].shape[0] - H
    wh = rnd(p['w'])[Din:]                                  # [H,4H]
    wci, wcf, wco = (p['wci'], p['wcf'], p['wco']) if use_peephole else (0.0, 0.0, 0.0)
    lens = np.asarray(lens)
    dgates = np.zeros((T, B, 4, H))
    dpeep = np.zeros((3, H))
    dh_rec = np.zeros((B, H)) if d_h_final is None else np.array(d_h_final, dtype=np.float64)
    dc_car = np.zeros((B, H)) if d_c_final is None else np.array(d_c_final, dtype=np.float64)
    rows = np.arange(B)
    for s in range((int(lens.max()) if B else 0) - 1, -1, -1):
        act = s < lens
        fr = np.where(act, (lens - 1 - s) if reverse else s, 0)
        hasp = act & (s > 0)
        frp = np.where(hasp, (lens - s) if reverse else s - 1, 0)
        gt = gates[fr, rows]
        i, g, f, o = gt[:, 0], gt[:, 1], gt[:, 2], gt[:, 3]
        c = cs[fr, rows]
        cprev = np.where(hasp[:, None], cs[frp, rows], 0.0)
        dh = dout[fr, rows] + dh_rec
        tc = np.tanh(c)
        d_o = dh * tc * o * (1 - o)
        dc = dc_car + dh * o * (1 - tc * tc) + d_o * wco
        d_g = dc * i * (1 - g * g)
        d_i = dc * g * i * (1 - i)
        d_f = dc * cprev * f * (1 - f)
        a = act[:, None]
        dG = np.where(a[:, :, None], np.stack([d_i, d_g, d_f, d_o], 1), 0.0)    # [B,4,H]
        dpeep[0] += (dG[:, 0] * cprev).sum(0)
        dpeep[1] += (dG[:, 2] * cprev).sum(0)
        dpeep[2] += (dG[:, 3] * c * a).sum(0)
        dGr = rnd(dG)
        ra, fa = rows[act], fr[act]
        dgates[fa, ra] = dGr[act]
        dc_car = np.where(a, dc * f + d_i * wci + d_f * wcf, dc_car)
        dh_rec = np.where(a, dGr.reshape(B, 4 * H) @ wh.T, dh_rec)
    db = dgates.sum((0, 1)).reshape(4 * H)
    return dict(dgates=dgates, dpeep=dpeep, db=db)


def layer_param_grads_np(x_tm, hout, dgates, lens, p, reverse=False, round_fn=None):
    """dW [Din+H,4H], dx [T,B,Din] from the gate gradients, as the host driver forms them
    (models/encoders/core/rnn_util.py): dW_x = x^T dG, dW_h = h_prev^T dG, dx = dG W_x^T."""
    import numpy as np
    rnd = round_fn if round_fn is not None else (lambda a: a)
    T, B, Din = x_tm.shape
    H = hout.shape[2]
    dg = dgates.reshape(T * B, 4 * H)
    hp = np.zeros_like(hout)
    if reverse:
        hp[:-1] = hout[1:]
    else:
        hp[1:] = hout[:-1]
    dw = np.concatenate([x_tm.reshape(T * B, Din).T @ dg, hp.reshape(T * B, H).T @ dg], 0)
    dx = (dg @ rnd(p['w'])[:Din].T).reshape(T, B, Din)
    return dw, dx


# --------------------------------------------------------------------------------------------------
# tf.contrib.rnn.LSTMCell with a projection layer (num_proj), the cell models/encoders/core/blstm.py:187-230 builds for
# lstm_impl == 'LSTMCell' (Sak et al. 2014, "LSTMP"): same gates as above (split order i, j, f, o; peepholes
# w_i_diag / w_f_diag on c_prev, w_o_diag on the new c; forget_bias added to f; cell clip before the output gate),
# then m = (sigmoid(o) * tanh(c)) @ projection/kernel [H, P]; the RECURRENT input and the emitted output are the
# projected m (kernel [(Din + P), 4H]), the state is (c [H], m [P]).  PINNED to the reference's Python LSTMCell as
# executed (tests/test_oracle_tfshim.py::test_tfshim_python_lstm_cell, ::test_tfshim_ctc_models[ctc_blstm_lstmcell_proj]).
def lstmp_cell(x, c_prev, m_prev, w, b, wci, wcf, wco, w_proj, forget_bias=1.0, cell_clip=0.0, use_peephole=True):
    c, h = lstm_block_cell(x, c_prev, m_prev, w, b, wci, wcf, wco, forget_bias, cell_clip, use_peephole,
                           clip_blocks_gradient=True)
    return c, h @ w_proj


def dynamic_rnn_p(x_tm, seq_len, p, reverse=False, drop_mask=None, forget_bias=1.0, cell_clip=0.0, use_peephole=True):
    """dynamic_rnn over one direction of projected cells; returns out [T,B,P], (c_final [B,H], m_final [B,P])."""
    T, B, _ = x_tm.shape
    H = p['b'].shape[0] // 4
    P = p['w_proj'].shape[1]
    if reverse:
        x_tm = reverse_sequence(x_tm, seq_len)
    c = x_tm.new_zeros(B, H)
    m = x_tm.new_zeros(B, P)
    outs = []
    for s in range(T):
        active = (s < seq_len).to(x_tm.dtype).unsqueeze(1)
        c_new, m_new = lstmp_cell(x_tm[s], c, m, p['w'], p['b'], p['wci'], p['wcf'], p['wco'], p['w_proj'],
                                  forget_bias, cell_clip, use_peephole)
        c = active * c_new + (1 - active) * c
        m = active * m_new + (1 - active) * m
        outs.append(m_new * active)
    out = torch.stack(outs, dim=0)
    if reverse:
        out = reverse_sequence(out, seq_len)
    if drop_mask is not None:
        out = out * drop_mask
    return out, (c, m)


def blstmp_encoder(inputs_bm, seq_len, layers, drop_masks=None, **kw):
    """BLSTMEncoder with lstm_impl='LSTMCell' and num_proj (blstm.py:187-230): layer outputs are [T,B,2P]."""
    x = inputs_bm.transpose(0, 1)
    final = None
    for li, (p_fw, p_bw) in enumerate(layers):
        m = drop_masks[li] if drop_masks is not None else (None, None)
        o_fw, st_fw = dynamic_rnn_p(x, seq_len, p_fw, False, m[0], **kw)
        o_bw, st_bw = dynamic_rnn_p(x, seq_len, p_bw, True, m[1], **kw)
        x = torch.cat([o_fw, o_bw], dim=2)
        final = (st_fw, st_bw)
    return x, final

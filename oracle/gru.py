"""Oracle (test infrastructure): CPU restatement of tf.contrib.rnn.GRUCell under
tf.nn.(bidirectional_)dynamic_rnn(sequence_length) as the reference's GRU encoders use it
(models/encoders/core/gru.py:58-76 GRUEncoder, :126-152 BGRUEncoder).

PINNED to TensorFlow's own known answers: core_rnn_cell_test.py::testGRUCell (kernels 0.5, x = [1, 1],
h = [0.1, 0.1] -> [0.175991, 0.175991]; x = [1, 1, 1], h = [0.1, 0.1] -> [0.156736, 0.156736]) -- these fix the cell
equations (TF 1.x):
    [r, u] = sigmoid([x, h] W_g + b_g)      b_g initialised to ONE
    c      = tanh([x, r * h] W_c + b_c)     the reset gate is applied BEFORE the candidate's matrix product
    h'     = u * h + (1 - u) * c
The sequence_length handling (state carried, output zero past the length, reverse_sequence for the backward
direction) is the one oracle.lstm.dynamic_rnn restates; the stacked encoders (scopes, MultiRNNCell, bidirectional concat)
are PINNED to the reference's own GRUEncoder / BGRUEncoder as executed (tests/test_oracle_tfshim.py::test_tfshim_ctc_models
[ctc_gru, ctc_bgru]) and cross-checked by finite differences in tests/test_oracle.py."""
import torch

from .lstm import reverse_sequence


def gru_cell(x, h, p):
    """x [B,D], h [B,H]; p = dict(wg [(D+H),2H], bg [2H], wc [(D+H),H], bc [H])."""
    H = h.shape[1]
    g = torch.sigmoid(torch.cat([x, h], 1) @ p['wg'] + p['bg'])
    r, u = g[:, :H], g[:, H:]
    c = torch.tanh(torch.cat([x, r * h], 1) @ p['wc'] + p['bc'])
    return u * h + (1 - u) * c


def dynamic_rnn(x_tm, seq_len, p, reverse=False, drop_mask=None):
    """x [T,B,D] time-major; returns out [T,B,H] (zero past seq_len, times the dropout mask), h_final [B,H]."""
    T, B, _ = x_tm.shape
    H = p['bc'].shape[0]
    if reverse:
        x_tm = reverse_sequence(x_tm, seq_len)
    h = x_tm.new_zeros(B, H)
    outs = []
    for t in range(T):
        active = (t < seq_len).to(x_tm.dtype).unsqueeze(1)
        hn = gru_cell(x_tm[t], h, p)
        h = active * hn + (1 - active) * h
        outs.append(active * hn)
    out = torch.stack(outs) if T else x_tm.new_zeros(0, B, H)
    if reverse:
        out = reverse_sequence(out, seq_len)
    if drop_mask is not None:
        out = out * drop_mask
    return out, h


def gru_encoder(x_tm, seq_len, layers, ndir, drop_masks=None):
    """layers: list of p (ndir = 1) or (p_fw, p_bw) (ndir = 2).  Returns out [T,B,ndir*H], final states of the last layer."""
    out = x_tm
    final = None
    for li, layer in enumerate(layers):
        dm = drop_masks[li] if drop_masks is not None else None
        if ndir == 2:
            H = layer[0]['bc'].shape[0]
            of, hf = dynamic_rnn(out, seq_len, layer[0], False, dm[:, :, :H] if dm is not None else None)
            ob, hb = dynamic_rnn(out, seq_len, layer[1], True, dm[:, :, H:] if dm is not None else None)
            out, final = torch.cat([of, ob], 2), (hf, hb)
        else:
            out, final = dynamic_rnn(out, seq_len, layer, False, dm)
    return out, final

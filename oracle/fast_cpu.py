"""Oracle, fast form (test infrastructure; used ONLY as bench.py's `cpu_baseline` "port"
leg and validated against oracle.model in tests/test_oracle.py).

A multi-threaded torch-CPU fp32 restatement of one training step of the reference's
TF1 CPU path for the BLSTM-CTC model: the same per-step structure TF1 executes --
one fused LSTMBlockCell evaluation per time step per direction per layer under a
dynamic_rnn loop (models/encoders/core/blstm.py:286-323), output FC
(models/ctc/ctc.py:216-224), CTC loss (ctc.py:289-298; here torch's CPU ctc_loss, which
tests pin to oracle.ctc), autograd backward, per-variable clip_by_norm
(models/model_base.py:148-152) and the optimizer update.  TensorFlow 1.x itself cannot be
installed in this image, so this is labelled "port", not "reference".
The input projection is hoisted out of the loop (one matmul over all T), which only makes
this baseline faster than TF1's per-step [x,h]W.
"""
import time

import numpy as np
import torch


def _cell(pre, c, wci, wcf, wco, H, clip):
    i = torch.sigmoid(pre[:, :H] + wci * c)
    g = torch.tanh(pre[:, H:2 * H])
    f = torch.sigmoid(pre[:, 2 * H:3 * H] + 1.0 + wcf * c)
    cn = g * i + c * f
    if clip and clip > 0:
        cn = cn + (cn.clamp(-clip, clip) - cn).detach()
    o = torch.sigmoid(pre[:, 3 * H:] + wco * cn)
    return cn, torch.tanh(cn) * o


def _direction(x_tm, seq_len, p, reverse, clip):
    T, B, D = x_tm.shape
    H = p['b'].shape[0] // 4
    if reverse:
        t = torch.arange(T).unsqueeze(1)
        L = seq_len.unsqueeze(0)
        src = torch.where(t < L, L - 1 - t, t)
        x_tm = x_tm[src, torch.arange(B).unsqueeze(0)]
    xproj = (x_tm.reshape(T * B, D) @ p['w'][:D] + p['b']).reshape(T, B, 4 * H)
    wh = p['w'][D:]
    c = x_tm.new_zeros(B, H)
    h = x_tm.new_zeros(B, H)
    outs = []
    act_all = (torch.arange(T).unsqueeze(1) < seq_len.unsqueeze(0)).to(x_tm.dtype).unsqueeze(2)
    for s in range(T):
        cn, hn = _cell(xproj[s] + h @ wh, c, p['wci'], p['wcf'], p['wco'], H, clip)
        a = act_all[s]
        c = a * cn + (1 - a) * c
        h = a * hn + (1 - a) * h
        outs.append(hn * a)
    out = torch.stack(outs, 0)
    if reverse:
        out = out[src, torch.arange(B).unsqueeze(0)]
    return out


class CpuBLSTMCTC(object):
    def __init__(self, state_dict, num_layers, cell_clip=0.0, clip_grad_norm=None, threads=None,
                 optimizer='momentum'):
        if threads:
            torch.set_num_threads(threads)
        assert optimizer in ('momentum', 'rmsprop')
        self.optimizer = optimizer
        self.L = num_layers
        self.clip = cell_clip
        self.clip_grad_norm = clip_grad_norm
        self.params = {k: torch.tensor(np.asarray(v), dtype=torch.float32).requires_grad_(True)
                       for k, v in state_dict.items()}
        self.mom = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.rms = {k: torch.ones_like(v) for k, v in self.params.items()}    # TF1 RMSProp: rms slot starts at 1

    def _layer_params(self, i, d):
        base = 'blstm_hidden%d/%s/lstm_cell' % (i, d)
        P = self.params
        return dict(w=P[base + '/kernel'], b=P[base + '/bias'], wci=P[base + '/w_i_diag'],
                    wcf=P[base + '/w_f_diag'], wco=P[base + '/w_o_diag'])

    def loss(self, x_btd, labels_list, seq_len):
        x = torch.as_tensor(x_btd, dtype=torch.float32).transpose(0, 1)
        sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
        for i in range(1, self.L + 1):
            fw = _direction(x, sl, self._layer_params(i, 'fw'), False, self.clip)
            bw = _direction(x, sl, self._layer_params(i, 'bw'), True, self.clip)
            x = torch.cat([fw, bw], 2)
        T, B, E = x.shape
        logits = (x.reshape(T * B, E) @ self.params['output/weights'] + self.params['output/biases'])
        logits = logits.reshape(T, B, -1)
        C = logits.shape[2]
        flat = torch.tensor(sum([list(l) for l in labels_list], []), dtype=torch.long)
        lens = torch.tensor([len(l) for l in labels_list], dtype=torch.long)
        losses = torch.nn.functional.ctc_loss(logits.log_softmax(2), flat, sl, lens, blank=C - 1,
                                              reduction='none', zero_infinity=True)
        return losses.mean(), logits

    def train_step(self, x_btd, labels_list, seq_len, lr=1e-3):
        """fwd + bwd + per-variable clip + update (momentum 0.9 or rmsprop); returns loss value."""
        for p in self.params.values():
            p.grad = None
        loss, _ = self.loss(x_btd, labels_list, seq_len)
        loss.backward()
        with torch.no_grad():
            for k, p in self.params.items():
                g = p.grad
                if self.clip_grad_norm:
                    n = g.norm()
                    g = g * (self.clip_grad_norm / torch.clamp(n, min=self.clip_grad_norm))
                if self.optimizer == 'rmsprop':   # tf.train.RMSPropOptimizer(decay 0.9, momentum 0, eps 1e-10)
                    self.rms[k].mul_(0.9).addcmul_(g, g, value=0.1)
                    p.sub_(lr * g / torch.sqrt(self.rms[k] + 1e-10))
                else:
                    self.mom[k].mul_(0.9).add_(g)
                    p.sub_(lr * self.mom[k])
        return float(loss.detach())


def time_train_steps(model, x, labels, seq_len, steps=1, warmup=0):
    for _ in range(warmup):
        model.train_step(x, labels, seq_len)
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(x, labels, seq_len)
    return (time.perf_counter() - t0) / max(steps, 1)

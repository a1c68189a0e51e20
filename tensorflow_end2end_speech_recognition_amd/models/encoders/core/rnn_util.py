"""Host driver of one (bi)directional LSTM layer on the HIP kernels.

Stands in for the per-layer body of lstmblockcell() in the reference
(models/encoders/core/blstm.py:281-323 / lstm.py:250-285):
LSTMBlockCell(forget_bias=1.0, clip_cell, use_peephole) fw (+ bw) under
DropoutWrapper(output_keep_prob) and (bidirectional_)dynamic_rnn(sequence_length).

Dataflow per layer (all device-side, one stream):
  forward : xproj = x W_x + b          one MFMA GEMM per direction over all T
            pack W_h -> fragments      (weights changed last step)
            lstm_fwd                   the serial recurrence kernel
            (dropout: mask on the emitted output only)
  backward: lstm_bwd                   BPTT -> dgates (pre-activation grads)
            dW_x = x^T dG, dW_h = h_prev^T dG, db = colsum(dG), dx = dG W_x^T   MFMA GEMMs
Variable names follow TF 1.3:
  <scope>/{fw,bw}/lstm_cell/{kernel,bias,w_i_diag,w_f_diag,w_o_diag}
"""
import numpy as np
import torch

from .... import ops
from ...._lib import ASR_BF16, ASR_F32

DIRS = ('fw', 'bw')


def declare_lstm_vars(store, scope, din, H, ndir, use_peephole, parameter_init, rng, cell_scope=None):
    """uniform(+-parameter_init) kernel and peepholes (variable_scope initializer,
    blstm.py:283-284), zero bias (LSTMBlockCell default)."""
    names = []
    for d in range(ndir):
        base = '%s/%s/lstm_cell' % (scope, DIRS[d]) if cell_scope is None else cell_scope
        u = lambda *s: rng.uniform(-parameter_init, parameter_init, size=s)
        store.declare(base + '/kernel', (din + H, 4 * H), u(din + H, 4 * H))
        store.declare(base + '/bias', (4 * H,), np.zeros(4 * H))
        if use_peephole:
            store.declare(base + '/w_i_diag', (H,), u(H))
            store.declare(base + '/w_f_diag', (H,), u(H))
            store.declare(base + '/w_o_diag', (H,), u(H))
        names.append(base)
    return names


class LSTMLayer(object):
    def __init__(self, store, bases, din, H, use_peephole, forget_bias=1.0, cell_clip=None):
        self.store = store
        self.bases = bases
        self.ndir = len(bases)
        self.din, self.H = din, H
        self.use_peephole = use_peephole
        self.forget_bias = forget_bias
        self.cell_clip = cell_clip
        self.ctx = None

    def _peep(self):
        if not self.use_peephole:
            return None
        st = self.store
        return torch.stack([torch.stack([st[b + '/w_i_diag'], st[b + '/w_f_diag'], st[b + '/w_o_diag']])
                            for b in self.bases]).contiguous()

    def prepare(self, device, dtype, T, B, keep_prob=1.0, is_training=True, rng_state=None, drop_mask=None):
        """Everything of a layer's forward that does not depend on its input: operand-dtype weight
        images (W_x transposed for the GEMM's fast path, W_h in MFMA fragment order for both
        passes), the peephole block and the dropout mask.  The encoder issues this for layer l+1
        on the side stream while layer l's recurrence runs."""
        st = self.store
        H, ndir, din = self.H, self.ndir, self.din
        wdt = torch.bfloat16 if dtype == ASR_BF16 else torch.float32
        whf = torch.empty((ndir, H * 4 * H), dtype=wdt, device=device)
        whb = torch.empty((ndir, H * 4 * H), dtype=wdt, device=device)
        wxT = torch.empty((ndir * 4 * H, din), dtype=wdt, device=device)      # both directions stacked
        bias = torch.empty((ndir * 4 * H,), dtype=torch.float32, device=device)
        wx_il = []
        for d, base in enumerate(self.bases):
            w = ops.lstm_prep_weights(st[base + '/kernel'], st[base + '/bias'], din, H, dtype,
                                      out=dict(wx_il=torch.empty((din, 4 * H), dtype=wdt, device=device),
                                               bias_il=bias[d * 4 * H:(d + 1) * 4 * H],
                                               pf=whf[d], pb=whb[d]))
            wx_il.append(w['wx_il'])
            # W_x is k-major ([Din, 4H]); the GEMM's fast path wants it reduction-contiguous
            ops.transpose2d(w['wx_il'], out=wxT[d * 4 * H:(d + 1) * 4 * H])
        mask = None
        if is_training and (drop_mask is not None or keep_prob < 1.0):
            if drop_mask is None:
                seed, offset = rng_state
                mask = ops.dropout_mask((T, B, ndir * H), keep_prob, seed, offset, device)
            else:
                mask = drop_mask
        # [Din, ndir*4H]: dx = dG [T*B, ndir*4H] . wx_cat^T is then ONE GEMM over both directions
        wx_cat = torch.cat(wx_il, dim=1) if ndir > 1 else wx_il[0]
        return dict(whf=whf, whb=whb, wxT=wxT, bias=bias, wx_cat=wx_cat, peep=self._peep(), mask=mask)

    def forward(self, x, seq_len, dtype, keep_prob=1.0, is_training=True, rng_state=None,
                drop_mask=None, save=True, prep=None):
        """x [T,B,din] in `dtype`; returns (out [T,B,ndir*H] in dtype, (c_final, h_final))."""
        T, B, din = x.shape
        H, ndir = self.H, self.ndir
        if prep is None:
            prep = self.prepare(x.device, dtype, T, B, keep_prob, is_training, rng_state, drop_mask)
        xproj = torch.empty((T, B, ndir * 4 * H), dtype=torch.float32, device=x.device)
        # x W_x + b for both directions in ONE GEMM (N = ndir*4H), written in the interleaved layout
        ops.gemm(x.view(T * B, din), prep['wxT'], transB=True, bias=prep['bias'],
                 out=xproj.view(T * B, ndir * 4 * H))
        gates, hout, cs, cf, hf = ops.lstm_fwd(xproj, prep['whf'], prep['peep'], seq_len, H, ndir, dtype,
                                               self.forget_bias, self.cell_clip or 0.0)
        out = hout
        mask = prep['mask']
        if mask is not None:
            out = ops.apply_mask(hout, mask)
        if save:
            self.ctx = dict(x=x, gates=gates, cs=cs, hout=hout, whb=prep['whb'], peep=prep['peep'],
                            seq_len=seq_len, dtype=dtype, mask=mask, wx_cat=prep['wx_cat'])
        return out, (cf, hf)

    def backward(self, dout, d_c_final=None, d_h_final=None, need_dx=True):
        """dout [T,B,ndir*H] fp32 -> dx [T,B,din] fp32 (or None).  Fills store.grad."""
        c = self.ctx
        st = self.store
        dtype = c['dtype']
        sh = st.shadow(dtype)
        x, hout = c['x'], c['hout']
        T, B, din = x.shape
        H, ndir = self.H, self.ndir
        if c['mask'] is not None:
            dout = ops.apply_mask(dout, c['mask'])
        dgates, dpeep = ops.lstm_bwd(dout, c['gates'], c['cs'], c['whb'], c['peep'], c['seq_len'], H,
                                     ndir, dtype, d_c_final, d_h_final, want_dpeep=True)
        x2d = x.view(T * B, din)
        h2d = hout.view(T * B, ndir * H)
        dg2d = dgates.view(T * B, ndir * 4 * H)
        dx = torch.empty((T, B, din), dtype=torch.float32, device=x.device) if need_dx else None
        if need_dx:   # the only result the layer below waits for: main stream, first
            ops.gemm(dg2d, c['wx_cat'], transB=True, out=dx.view(T * B, din))
        # weight gradients: side stream, concurrent with the BPTT kernel of the layer below
        # (joined in the model's backward before clipping)
        dw_il = torch.empty((ndir, din + H, 4 * H), dtype=torch.float32, device=x.device)   # interleaved cols
        with ops.side_lane(x.device, keep=(x, hout, dgates, dpeep, dw_il)):
            for d, base in enumerate(self.bases):
                dg = dg2d[:, d * 4 * H:(d + 1) * 4 * H]
                ops.gemm(x2d, dg, transA=True, out=dw_il[d, :din])
                if T > 1:
                    if d == 0:   # forward direction: h_prev(t) = h(t-1)
                        ops.gemm(h2d[:(T - 1) * B, d * H:(d + 1) * H], dg[B:], transA=True, out=dw_il[d, din:])
                    else:        # backward direction: h_prev(t) = h(t+1) (zero beyond len-1)
                        ops.gemm(h2d[B:, d * H:(d + 1) * H], dg[:(T - 1) * B], transA=True, out=dw_il[d, din:])
                else:
                    dw_il[d, din:].zero_()
                ops.gate_deinterleave(dw_il[d], st.g(base + '/kernel'), H)
                st.g(base + '/bias').copy_(dpeep[d, 3:7].reshape(-1))   # bias grad accumulated inside BPTT
                if self.use_peephole:
                    st.g(base + '/w_i_diag').copy_(dpeep[d, 0])
                    st.g(base + '/w_f_diag').copy_(dpeep[d, 1])
                    st.g(base + '/w_o_diag').copy_(dpeep[d, 2])
        self.ctx = None
        return dx

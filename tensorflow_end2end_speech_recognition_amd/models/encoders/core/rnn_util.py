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
import os as _os

import numpy as np
import torch

from .... import ops
from ...._lib import ASR_BF16, ASR_F32

DIRS = ('fw', 'bw')
# side streams the weight-gradient GEMMs of a layer are spread over: one per direction (one lane for both measures the
# same step time, 12.39 ms, but leaves a longer tail after the last BPTT kernel)
DW_LANES = 2
BG_WGS = int(_os.environ.get('ASR_BG_WGS', '32'))   # workgroups of a weight-gradient GEMM that runs beside a BPTT kernel


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
        self.grad_event = None

    def _vars(self, view):
        """Per direction (kernel, bias, w_i_diag, w_f_diag, w_o_diag) through `view` (store[...] or store.g)."""
        rows = []
        for b in self.bases:
            row = [view(b + '/kernel'), view(b + '/bias')]
            row += [view(b + '/w_i_diag'), view(b + '/w_f_diag'), view(b + '/w_o_diag')] if self.use_peephole \
                else [None, None, None]
            rows.append(tuple(row))
        return rows

    def prepare(self, device, dtype, T, B, keep_prob=1.0, is_training=True, rng_state=None, drop_mask=None,
                ldk=None):
        """Everything of a layer's forward that does not depend on its input: operand-dtype weight
        images (W_x transposed + zero-padded to the input's row width `ldk` for the GEMM's fast path, W_x
        with both directions side by side for dx, W_h in MFMA fragment order for both passes, the
        peephole block: ONE launch) and the dropout mask.  The encoder issues this for every layer on
        the side stream at the start of the step, under the first recurrence kernels."""
        w = ops.lstm_prep_layer(self._vars(self.store.__getitem__), self.din, self.H, dtype, ldk=ldk)
        w['mask'] = self.make_mask(device, T, B, keep_prob, is_training, rng_state, drop_mask)
        return w

    def make_mask(self, device, T, B, keep_prob=1.0, is_training=True, rng_state=None, drop_mask=None):
        """Dropout mask of the layer's output, or None: a caller-supplied tensor ([T,B,ndir*H] fp32, 1/keep_prob or 0),
        or -- the normal case -- the descriptor (keep_prob, seed, offset) of the mask ops.dropout_mask would generate
        with those arguments.  The kernels that apply it (asr_dropout_apply, the dx GEMM's epilogue: asr_gemm_drop)
        form it from the generator's counter, so no [T,B,ndir*H] fp32 tensor is written beside the first recurrence
        (5 x 25 MB at cfg B: the first layer's forward kernel ran 85 us longer than the others) or read back."""
        if not (is_training and (drop_mask is not None or keep_prob < 1.0)):
            return None
        if drop_mask is not None:
            return drop_mask
        seed, offset = rng_state
        return (float(keep_prob), int(seed), int(offset))

    @staticmethod
    def _masked(t, mask):
        return ops.dropout_apply(t, *mask) if isinstance(mask, tuple) else ops.apply_mask(t, mask)

    def forward(self, x, seq_len, dtype, keep_prob=1.0, is_training=True, rng_state=None,
                drop_mask=None, save=True, prep=None, mask_event=None):
        """x [T,B,ldk] in `dtype` (ldk >= din: columns din.. are zero padding); returns
        (out [T,B,ndir*H] in dtype, (c_final, h_final))."""
        T, B, ldk = x.shape
        H, ndir = self.H, self.ndir
        if prep is None:
            prep = self.prepare(x.device, dtype, T, B, keep_prob, is_training, rng_state, drop_mask, ldk=ldk)
        xproj = torch.empty((T, B, ndir * 4 * H), dtype=torch.float32, device=x.device)
        # x W_x + b for both directions in ONE GEMM (N = ndir*4H), written in the interleaved layout
        ops.gemm(x.view(T * B, ldk), prep['wxT'], transB=True, bias=prep['bias'],
                 out=xproj.view(T * B, ndir * 4 * H))
        gates, hout, cs, cf, hf = ops.lstm_fwd(xproj, prep['whf'], prep['peep'], seq_len, H, ndir, dtype,
                                               self.forget_bias, self.cell_clip or 0.0)
        out = hout
        mask = prep['mask']
        if mask is not None:
            if not isinstance(mask, tuple):
                ops.wait_event(mask_event)   # a mask tensor may have been produced on the side stream
            out = self._masked(hout, mask)
        if save:
            self.ctx = dict(x=x, gates=gates, cs=cs, hout=hout, whb=prep['whb'], peep=prep['peep'],
                            seq_len=seq_len, dtype=dtype, mask=mask, wx_cat=prep['wx_cat'])
        return out, (cf, hf)

    def backward(self, dout, d_c_final=None, d_h_final=None, need_dx=True, dout_masked=False, dx_mask=None,
                 background=False, warm=None):
        """dout [T,B,ndir*H] fp32 -> dx [T,B,din] fp32 (or None).  Fills store.grad.
        dout_masked: the caller has already multiplied dout with this layer's dropout mask.
        dx_mask: dropout mask [T,B,din] of the layer BELOW: dx comes back already multiplied with it (in the epilogue
        of the dx GEMM), i.e. ready to be passed to that layer's backward with dout_masked=True.
        background: another recurrence kernel follows (the layer below's BPTT): the weight-gradient GEMMs then have a
        millisecond to finish beside it and run on few workgroups without split-K slabs, whose traffic would slow it."""
        c = self.ctx
        st = self.store
        dtype = c['dtype']
        x, hout = c['x'], c['hout']
        T, B, ldk = x.shape
        din, H, ndir = self.din, self.H, self.ndir
        if c['mask'] is not None and not dout_masked:
            dout = self._masked(dout, c['mask'])
        dgates, dpeep = ops.lstm_bwd(dout, c['gates'], c['cs'], c['whb'], c['peep'], c['seq_len'], H,
                                     ndir, dtype, d_c_final, d_h_final, want_dpeep=True)
        self.warm_event = None
        if warm:
            # warm: the saved activations of the layer BELOW, read once on a side lane while this layer's dx product runs
            # (after this BPTT kernel, before the next one): its BPTT kernel then finds them in the memory-side cache
            with ops.side_lane(dout.device, keep=tuple(warm), lane=2):
                for t in warm:
                    ops.touch(t)
                self.warm_event = ops.stream_event()
        x2d = x.view(T * B, ldk)[:, :din]
        h2d = hout.view(T * B, ndir * H)
        dg2d = dgates.view(T * B, ndir * 4 * H)
        dx = torch.empty((T, B, din), dtype=torch.float32, device=x.device) if need_dx else None
        if need_dx:   # the only result the layer below waits for: main stream, first
            if isinstance(dx_mask, tuple):
                ops.gemm(dg2d, c['wx_cat'], transB=True, out=dx.view(T * B, din), drop=dx_mask)
            else:
                ops.gemm(dg2d, c['wx_cat'], transB=True, out=dx.view(T * B, din),
                         mul=dx_mask.view(T * B, din) if dx_mask is not None else None)
        # weight gradients: side streams (one per direction), concurrent with the BPTT kernel of the layer
        # below (joined in the model's backward before clipping)
        dw_il = torch.empty((ndir, din + H, 4 * H), dtype=torch.float32, device=x.device)   # interleaved cols
        ops.set_side_gemm_workgroups(x.device, BG_WGS if background else 0)
        done = []
        for d in range(ndir):
            with ops.side_lane(x.device, keep=(x, hout, dgates, dpeep, dw_il), lane=1 + (d % DW_LANES)):
                dg = dg2d[:, d * 4 * H:(d + 1) * 4 * H]
                ops.gemm(x2d, dg, transA=True, out=dw_il[d, :din])
                if T > 1:
                    if d == 0:   # forward direction: h_prev(t) = h(t-1)
                        ops.gemm(h2d[:(T - 1) * B, d * H:(d + 1) * H], dg[B:], transA=True, out=dw_il[d, din:])
                    else:        # backward direction: h_prev(t) = h(t+1) (zero beyond len-1)
                        ops.gemm(h2d[B:, d * H:(d + 1) * H], dg[:(T - 1) * B], transA=True, out=dw_il[d, din:])
                else:
                    dw_il[d, din:].zero_()
                if d % DW_LANES > 0:
                    done.append(ops.stream_event())
        with ops.side_lane(x.device, lane=1):
            for ev in done:
                ops.wait_event(ev)
            # interleaved columns -> TF's gate-major kernel gradient; bias / peephole gradients (accumulated
            # inside the BPTT kernel) to their variables: one launch for the layer
            ops.lstm_grad_finish(self._vars(st.g), dw_il, dpeep, H)
            # every gradient of this layer is complete at this point of side lane 1 (the data-parallel step hangs
            # the layer's clip + all-reduce on it while the layers below are still in their BPTT)
            self.grad_event = ops.stream_event()
        self.ctx = None
        return dx

    def var_names(self):
        """This layer's variables, in declaration order (one contiguous run of the ParamStore)."""
        names = []
        for b in self.bases:
            names += [b + '/kernel', b + '/bias']
            if self.use_peephole:
                names += [b + '/w_i_diag', b + '/w_f_diag', b + '/w_o_diag']
        return names

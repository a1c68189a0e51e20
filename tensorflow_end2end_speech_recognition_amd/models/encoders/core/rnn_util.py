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
# (round 6: the bottom layer's two recurrent products moved off these lanes -- a third lane and the main stream, all four
# products side by side behind the last BPTT kernel -- measured: last BPTT end -> grad_finish 127 us against 114 us as it is,
# step 8.911 against 8.899 ms; the four split-K GEMMs already fill the chip two at a time.  Not kept.)
# one main-stream marker per BPTT layer for all its side lanes, one wait for all weight images (A-B: ASR_FORK_ONCE=0
# restores a marker per lane entry and a wait per layer)
FORK_ONCE = _os.environ.get('ASR_FORK_ONCE', '1') != '0'
BG_WGS = int(_os.environ.get('ASR_BG_WGS', '128'))   # workgroups of a weight-gradient GEMM that runs beside a BPTT kernel
# a dx product of at least this many flops keeps the chip to itself: the layer's weight-gradient lanes start BEHIND it (one
# more main-stream marker) instead of behind the BPTT kernel.  Timeline of the cfg-C-shaped step (4 x 512, 104 k frames,
# profiles/r05_cfgC_timeline.md): the 874 GFLOP dx product took 2.1 - 2.4 ms beside the two W_x gradient GEMMs (1.04 alone)
# and the layer below cannot start before it, while its BPTT kernel then left half the chip idle for 2 ms.
# ASR_DW_AFTER_DX=0: never, =1: always.
DW_AFTER_DX_FLOPS = {'0': float('inf'), '1': 0.0}.get(_os.environ.get('ASR_DW_AFTER_DX', ''), 2e11)


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
                 background=False, warm=None, acc=None, finish=True):
        """dout [T,B,ndir*H] fp32 -> dx [T,B,din] fp32 (or None).  Fills store.grad.
        acc / finish (the encoder's two half-batch pipelines, blstm.py): finish=False leaves the weight gradients of THIS
        part of the batch in self.acc = dict(dw_il, dpeep) instead of the variables; the twin layer that handles the other
        part takes that dict as `acc`, adds its own products to it (GEMMs with accumulate on the same side lanes, i.e.
        behind the first part's in lane order: a fixed summation order) and writes the variables.
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
        # ONE marker on the main stream behind the BPTT kernel: every side lane of this layer starts there
        fork = ops.stream_event() if FORK_ONCE else None
        if warm:
            # warm: the saved activations of the layer BELOW, read once on a side lane while this layer's dx product runs
            # (after this BPTT kernel, before the next one): its BPTT kernel then finds them in the memory-side cache
            with ops.side_lane(dout.device, keep=tuple(warm), lane=2, after=fork):
                for t in warm:
                    ops.touch(t)
                if not FORK_ONCE:
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
            if FORK_ONCE and background and 2.0 * T * B * din * ndir * 4 * H >= DW_AFTER_DX_FLOPS:
                fork = ops.stream_event()
        # weight gradients: side streams (one per direction), concurrent with the BPTT kernel of the layer
        # below (joined in the model's backward before clipping)
        add = acc is not None
        dw_il = acc['dw_il'] if add else \
            torch.empty((ndir, din + H, 4 * H), dtype=torch.float32, device=x.device)   # interleaved cols
        ops.set_side_gemm_workgroups(x.device, BG_WGS if background else 0)
        done = []
        for d in range(ndir):
            with ops.side_lane(x.device, keep=(x, hout, dgates, dpeep, dw_il), lane=1 + (d % DW_LANES), after=fork):
                dg = dg2d[:, d * 4 * H:(d + 1) * 4 * H]
                ops.gemm(x2d, dg, transA=True, out=dw_il[d, :din], accumulate=add)
                if T > 1:
                    if d == 0:   # forward direction: h_prev(t) = h(t-1)
                        ops.gemm(h2d[:(T - 1) * B, d * H:(d + 1) * H], dg[B:], transA=True, out=dw_il[d, din:],
                                 accumulate=add)
                    else:        # backward direction: h_prev(t) = h(t+1) (zero beyond len-1)
                        ops.gemm(h2d[B:, d * H:(d + 1) * H], dg[:(T - 1) * B], transA=True, out=dw_il[d, din:],
                                 accumulate=add)
                elif not add:
                    dw_il[d, din:].zero_()
                if d % DW_LANES > 0:
                    done.append(ops.stream_event())
        if not finish:
            # the other part of the batch finishes the layer (its GEMMs follow these in the lanes' order)
            self.acc = dict(dw_il=dw_il, dpeep=dpeep)
            self.ctx = None
            return dx
        with ops.side_lane(x.device, keep=(dpeep,), lane=1, after=fork):
            for ev in done:
                ops.wait_event(ev)
            if add:      # bias / peephole sums of the two parts (lane 1 already stands behind the first part's BPTT)
                dpeep = acc['dpeep'] + dpeep
                ops.keep_on_lane(x.device, 1, (dpeep,))
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


# ----------------------------------------------------------------------------------------------------------------------
# tf.contrib.rnn.LSTMCell WITH A PROJECTION LAYER (num_proj) -- the cell the reference builds for lstm_impl == 'LSTMCell'
# (models/encoders/core/blstm.py:187-230, lstm.py:117-180; Sak et al. 2014): the recurrent input and the emitted output
# are m = (o * tanh(c)) @ projection/kernel [H, P], so the layer's recurrence is  pre = x W_x + b + m_prev W_h  with
# W_h [P, 4H], and its outputs are [T, B, ndir * P].
# The multi-CU recurrence kernels are built around an H x 4H recurrent matrix -- and the projected cell HAS one: the state
# that feeds back is m = h W_p, so  m_prev W_h = h_prev (W_p W_h)  and the recurrence over the UNPROJECTED h = o * tanh(c)
# is the plain cell's with W' = W_p W_h [H, 4H] (one small GEMM per step of training).  The fused path (round 6, the
# default wherever asr_lstm_fwd takes the width: H in 64/128/192/256/320/512) therefore runs asr_lstm_fwd / asr_lstm_bwd_ex
# -- the cluster kernels at 128/256/320/512 -- on W', and everything else is batched over all T:
#   forward : x W_x + b,  the recurrence,  m = h W_p (the layer's output; padded frames are zero because h is)
#   backward: dh = dout W_p^T,  BPTT on W'^T (clip_no_grad: LSTMCell clamps with tf.clip_by_value, a clamped state passes
#             nothing back -- asr_lstm_bwd_ex),  Z = dG W_h^T (the part of dm that came back through the recurrence),
#             dW_x = x^T dG,  dW_h = m_prev^T dG,  dW_p = h^T dout + h^T Z_next,  dx = dG W_x^T,  bias / peepholes from the
#             BPTT kernel's column sums.
# (h W_p) W_h and h (W_p W_h) differ by fp32 rounding only (the parity test holds the same 1e-4 / 2e-3 bars as before).
# Operand dtype: fp32 (the three-term cluster kernels) for an fp32 model; for a bf16 model the bf16 recurrence kernels -- the
# headline's clusters, with the gradient-blocking clip as their own instantiations -- and bf16 operands in every batched
# product, W' rounded once like every other weight image, the emitted m rounded as LSTMLayer's h is; the layer's inputs,
# outputs and gradients are fp32 at its boundary either way (blstm 5 x 256 / projection 128 on the headline batch: 20.0 ms per
# step in fp32, 9.4 in bf16; loss 4e-6 apart).
# Other widths -- and ASR_LSTMP_FUSED=0, the A/B switch the tests use to hold the two paths against each other -- run the
# recurrence step by step on the generic kernels (per step: one skinny MFMA product for the recurrent term,
# asr_lstm_cell_fwd, one for the projection; backwards the mirror image), with everything that does not feed back batched
# into single GEMMs as well.
def declare_lstmp_vars(store, scope, din, H, P, ndir, use_peephole, parameter_init, rng, cell_scope=None):
    names = []
    for d in range(ndir):
        base = '%s/%s/lstm_cell' % (scope, DIRS[d]) if cell_scope is None else cell_scope
        u = lambda *s: rng.uniform(-parameter_init, parameter_init, size=s)
        store.declare(base + '/kernel', (din + P, 4 * H), u(din + P, 4 * H))
        store.declare(base + '/bias', (4 * H,), np.zeros(4 * H))
        if use_peephole:
            store.declare(base + '/w_f_diag', (H,), u(H))
            store.declare(base + '/w_i_diag', (H,), u(H))
            store.declare(base + '/w_o_diag', (H,), u(H))
        store.declare(base + '/projection/kernel', (H, P), u(H, P))
        names.append(base)
    return names


class LSTMPLayer(object):
    def __init__(self, store, bases, din, H, P, use_peephole, forget_bias=1.0, cell_clip=None, dtype=ASR_F32):
        self.store, self.bases = store, bases
        self.ndir = len(bases)
        self.din, self.H, self.P = din, H, P
        # operand dtype of the fused path's GEMMs and recurrences (the layer's inputs, outputs and gradients stay fp32 at its
        # boundary; the step-by-step path is fp32 throughout)
        self.dtype = dtype
        self.use_peephole = use_peephole
        self.forget_bias, self.cell_clip = forget_bias, cell_clip
        self.ctx = None
        self.grad_event = None

    def var_names(self):
        names = []
        for b in self.bases:
            names += [b + '/kernel', b + '/bias']
            if self.use_peephole:
                names += [b + '/w_f_diag', b + '/w_i_diag', b + '/w_o_diag']
            names.append(b + '/projection/kernel')
        return names

    def _peep(self, b):
        st = self.store
        if not self.use_peephole:
            return None
        return torch.stack([st[b + '/w_i_diag'], st[b + '/w_f_diag'], st[b + '/w_o_diag']]).contiguous()

    @staticmethod
    def _frames(seq_len, T, reverse):
        """[T,B] frame index each row works on at recurrence step s (tf.reverse_sequence on the valid prefix for the
        backward direction; rows past their length stay on frame s) and the live mask [T,B] fp32."""
        B = seq_len.shape[0]
        s = torch.arange(T, device=seq_len.device).unsqueeze(1)
        L = seq_len.to(torch.int64).unsqueeze(0)
        live = s < L
        src = torch.where(live, L - 1 - s, s) if reverse else s.expand(T, B)
        return src.contiguous(), live.to(torch.float32).contiguous()

    def _vars(self, view):
        rows = []
        for b in self.bases:
            row = [view(b + '/kernel'), view(b + '/bias')]
            row += [view(b + '/w_i_diag'), view(b + '/w_f_diag'), view(b + '/w_o_diag')] if self.use_peephole \
                else [None, None, None]
            rows.append(tuple(row))
        return rows

    def fused(self):
        """The recurrence on W' = W_p W_h through asr_lstm_fwd / asr_lstm_bwd_ex (see the comment above the class)."""
        return _os.environ.get('ASR_LSTMP_FUSED', '1') != '0' and ops.lstm_units_supported(self.H)

    def operand_dtype(self):
        """bf16 operands (a bf16 model: the headline's recurrence kernels, W' = W_p W_h rounded once like every other weight
        image) when the slices the batched products address are 16-byte aligned in bf16; fp32 otherwise and under
        ASR_LSTMP_BF16=0 (A/B)."""
        if self.dtype == ASR_BF16 and self.din % 8 == 0 and self.P % 8 == 0 and \
                _os.environ.get('ASR_LSTMP_BF16', '1') != '0':
            return ASR_BF16
        return ASR_F32

    def _forward_fused(self, x, seq_len, mask, save):
        st = self.store
        T, B, din = x.shape
        H, P, ndir = self.H, self.P, self.ndir
        dev = x.device
        Bp = (B + 15) // 16 * 16
        if Bp != B:   # the recurrence kernels work on 16-utterance tiles: rows of length 0 fill the last one
            x = torch.cat([x, x.new_zeros((T, Bp - B, din))], 1)
            seq_len = torch.cat([seq_len, seq_len.new_zeros((Bp - B,))])
        dt = self.operand_dtype()
        lo = dt != ASR_F32
        x = ops.cast_from_f32(x.contiguous(), dt) if lo else x.contiguous()
        # the layer as a plain cell over the input [x, m_prev-as-input-columns]: kernel rows [W_x; W_h; W'] -- the prep
        # launch then also yields W_h in the interleaved gate layout the BPTT kernel's dG has (columns din.. of wx_cat)
        rows = []
        for b, v in zip(self.bases, self._vars(st.__getitem__)):
            kf = torch.empty((din + P + H, 4 * H), dtype=torch.float32, device=dev)
            kf[:din + P].copy_(v[0])
            ops.gemm(st[b + '/projection/kernel'], v[0][din:], out=kf[din + P:])                 # W' = W_p W_h
            rows.append((kf,) + v[1:])
        prep = ops.lstm_prep_layer(rows, din + P, H, dt)
        xproj = torch.empty((T, Bp, ndir * 4 * H), dtype=torch.float32, device=dev)
        ops.gemm(x.view(T * Bp, din), prep['wxT'][:, :din], transB=True, bias=prep['bias'],
                 out=xproj.view(T * Bp, ndir * 4 * H))
        gates, hout, cs, cf, hf = ops.lstm_fwd(xproj, prep['whf'], prep['peep'], seq_len, H, ndir, dt,
                                               self.forget_bias, self.cell_clip or 0.0)
        m = torch.empty((T, Bp, ndir * P), dtype=hout.dtype, device=dev)      # (bf16 operands: rounded as LSTMLayer's h is)
        finals, wps = [], []
        for d, b in enumerate(self.bases):
            wp = st[b + '/projection/kernel']
            wps.append(ops.cast_from_f32(wp, dt) if lo else wp)
            ops.gemm(hout.view(T * Bp, ndir * H)[:, d * H:(d + 1) * H], wps[d], out=m.view(T * Bp, ndir * P)[:, d * P:(d + 1) * P])
            finals.append((cf[d, :B], ops.gemm(hf[d], wp)[:B]))
        mo = ops.cast_to_f32(m) if lo else m
        out = mo[:, :B].contiguous() if Bp != B else mo
        res = out if mask is None else ops.apply_mask(out, mask)
        if save:
            self.ctx = dict(fused=True, x=x, m=m, hout=hout, hf=hf, gates=gates, cs=cs, whb=prep['whb'], peep=prep['peep'],
                            wx_cat=prep['wx_cat'], seq_len=seq_len, mask=mask, batch=B, dt=dt, wps=wps)
        return res, finals

    def _backward_fused(self, dout, d_final, need_dx):
        c, st = self.ctx, self.store
        x, m, hout = c['x'], c['m'], c['hout']
        T, Bp, din = x.shape
        H, P, ndir, B = self.H, self.P, self.ndir, c['batch']
        dev = x.device
        if c['mask'] is not None:
            dout = ops.apply_mask(dout.contiguous(), c['mask'])
        if Bp != B:
            dout = torch.cat([dout, dout.new_zeros((T, Bp - B, ndir * P))], 1)
        dt, wps = c['dt'], c['wps']
        lo = dt != ASR_F32
        dout = ops.cast_from_f32(dout.contiguous(), dt) if lo else dout.contiguous()
        do2d, h2d, m2d = dout.view(T * Bp, ndir * P), hout.view(T * Bp, ndir * H), m.view(T * Bp, ndir * P)
        dh = torch.empty((T, Bp, ndir * H), dtype=torch.float32, device=dev)
        dcf = dhf = None
        if d_final is not None and any(f is not None for f in d_final):
            dcf = torch.zeros((ndir, Bp, H), dtype=torch.float32, device=dev)
            dhf = torch.zeros((ndir, Bp, H), dtype=torch.float32, device=dev)
        for d, b in enumerate(self.bases):
            wp = st[b + '/projection/kernel']
            ops.gemm(do2d[:, d * P:(d + 1) * P], wps[d], transB=True, out=dh.view(T * Bp, ndir * H)[:, d * H:(d + 1) * H])
            if dcf is not None and d_final[d] is not None:
                dcf[d, :B].copy_(d_final[d][0])
                ops.gemm(d_final[d][1].contiguous(), wp, transB=True, out=dhf[d, :B])
        dgates, dpeep = ops.lstm_bwd(dh, c['gates'], c['cs'], c['whb'], c['peep'], c['seq_len'], H, ndir, dt,
                                     dcf, dhf, want_dpeep=True, clip_no_grad=self.cell_clip or 0.0)
        dg2d = dgates.view(T * Bp, ndir * 4 * H)
        wx_cat = c['wx_cat']                                       # [din + P, ndir * 4H], dG's column order
        # ONE marker behind the BPTT kernel; the main stream carries only what the layer below waits for (dx), the weight
        # gradients run on side lanes (one per direction) beside it and beside the layer below's recurrence -- the encoder's
        # backward joins the lanes before the clip (as LSTMLayer / GRULayer)
        fork = ops.stream_event()
        dx = None
        if need_dx:
            dx = ops.gemm(dg2d, wx_cat[:din], transB=True, out_dtype=torch.float32).view(T, Bp, din)
            if Bp != B:
                dx = dx[:, :B].contiguous()
        dw_il = torch.empty((ndir, din + P, 4 * H), dtype=torch.float32, device=dev)
        x2d = x.view(T * Bp, din)
        lo, hi = slice(0, (T - 1) * Bp), slice(Bp, T * Bp)
        held = (x, m, hout, dout, dgates, dpeep, dw_il, wx_cat, c['hf']) + tuple(wps) + \
            tuple(t for f in (d_final or ()) if f is not None for t in f)
        done = []
        for d, b in enumerate(self.bases):
            with ops.side_lane(dev, keep=held, lane=1 + (d % 2), after=fork):
                dg = dg2d[:, d * 4 * H:(d + 1) * 4 * H]
                hd, md = h2d[:, d * H:(d + 1) * H], m2d[:, d * P:(d + 1) * P]
                gp = st.g(b + '/projection/kernel')
                ops.gemm(x2d, dg, transA=True, out=dw_il[d, :din])
                ops.gemm(hd, do2d[:, d * P:(d + 1) * P], transA=True, out=gp)    # dW_p: the part through the emitted output
                if dcf is not None and d_final[d] is not None:                   # ... and through the final m
                    ops.gemm(c['hf'][d, :B], d_final[d][1].contiguous(), transA=True, out=gp, accumulate=True)
                if T > 1:
                    # the step after frame t is frame t+1 (forward direction) / t-1 (backward); dG and h are zero at padded
                    # frames, so the shifted products need no mask
                    prev, nxt = (lo, hi) if d == 0 else (hi, lo)
                    z = ops.gemm(dg, wx_cat[din:, d * 4 * H:(d + 1) * 4 * H], transB=True)   # Z = dG W_h^T  [T*Bp, P] (operand dtype)
                    ops.gemm(md[prev], dg[nxt], transA=True, out=dw_il[d, din:])             # dW_h = m_prev^T dG
                    ops.gemm(hd[prev], z[nxt], transA=True, out=gp, accumulate=True)         # dW_p += h^T Z_next
                else:
                    dw_il[d, din:].zero_()
                if d % 2 > 0:
                    done.append(ops.stream_event())
        with ops.side_lane(dev, lane=1, after=fork):
            for ev in done:
                ops.wait_event(ev)
            ops.lstm_grad_finish(self._vars(st.g), dw_il, dpeep, H)
            self.grad_event = ops.stream_event()
        self.ctx = None
        return dx

    def forward(self, x, seq_len, mask=None, save=True):
        """x [T,B,din] fp32 time-major -> (out [T,B,ndir*P], per direction final (c [B,H], m [B,P]))."""
        if self.fused():
            return self._forward_fused(x, seq_len, mask, save)
        st = self.store
        T, B, din = x.shape
        H, P, ndir = self.H, self.P, self.ndir
        dev = x.device
        out = torch.zeros((T, B, ndir * P), dtype=torch.float32, device=dev)
        bidx = torch.arange(B, device=dev).unsqueeze(0)
        saved, finals = [], []
        for d, b in enumerate(self.bases):
            w = st[b + '/kernel']
            wx, wh, wp = w[:din], w[din:], st[b + '/projection/kernel']
            peep = self._peep(b)
            src, live = self._frames(seq_len, T, d == 1)
            pre = ops.gemm(x.view(T * B, din), wx, bias=st[b + '/bias']).view(T, B, 4 * H)
            pre = pre[src, bidx].contiguous()                 # step-major: row b of step s holds its own frame
            c = torch.zeros((B, H), dtype=torch.float32, device=dev)
            m = torch.zeros((B, P), dtype=torch.float32, device=dev)
            hdummy = torch.zeros((B, H), dtype=torch.float32, device=dev)
            gates_all = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
            craw_all = torch.empty((T, B, H), dtype=torch.float32, device=dev)
            cprev_all = torch.empty((T, B, H), dtype=torch.float32, device=dev)
            hraw_all = torch.empty((T, B, H), dtype=torch.float32, device=dev)
            mprev_all = torch.empty((T, B, P), dtype=torch.float32, device=dev)
            mstep = torch.zeros((T, B, P), dtype=torch.float32, device=dev)
            for s in range(T):
                cprev_all[s].copy_(c)
                mprev_all[s].copy_(m)
                ops.gemm(m, wh, out=pre[s], accumulate=True)                           # + m_prev W_h
                g, c_raw, c, _, h_raw = ops.lstm_cell_fwd(pre[s], c, hdummy, peep, live[s], self.forget_bias,
                                                          self.cell_clip or 0.0)
                gates_all[s].copy_(g)
                craw_all[s].copy_(c_raw)
                hraw_all[s].copy_(h_raw)
                m_new = ops.gemm(h_raw, wp)                                            # projection
                lv = live[s].unsqueeze(1)
                m = m_new * lv + m * (1.0 - lv)               # dynamic_rnn: finished rows keep their state
                mstep[s].copy_(m_new * lv)                    #              and emit zeros
            # step-major -> frame-major (reverse_sequence back); padded frames are zero already
            o = torch.zeros((T, B, P), dtype=torch.float32, device=dev)
            o[src, bidx] = mstep
            o = o * (torch.arange(T, device=dev).unsqueeze(1) < seq_len.to(torch.int64).unsqueeze(0)).unsqueeze(2)
            out[:, :, d * P:(d + 1) * P] = o
            finals.append((c, m))
            saved.append(dict(src=src, live=live, gates=gates_all, craw=craw_all, cprev=cprev_all, hraw=hraw_all,
                              mprev=mprev_all, peep=peep))
        res = out if mask is None else ops.apply_mask(out, mask)
        if save:
            self.ctx = dict(x=x, saved=saved, mask=mask, seq_len=seq_len)
        return res, finals

    def backward(self, dout, d_final=None, need_dx=True):
        """dout [T,B,ndir*P] fp32 -> dx [T,B,din] or None.  d_final: per direction (dc [B,H], dm [B,P]) or None."""
        if self.ctx.get('fused'):
            return self._backward_fused(dout, d_final, need_dx)
        c, st = self.ctx, self.store
        x = c['x']
        T, B, din = x.shape
        H, P, ndir = self.H, self.P, self.ndir
        dev = x.device
        if c['mask'] is not None:
            dout = ops.apply_mask(dout.contiguous(), c['mask'])
        bidx = torch.arange(B, device=dev).unsqueeze(0)
        dx = torch.zeros((T, B, din), dtype=torch.float32, device=dev) if need_dx else None
        for d, b in enumerate(self.bases):
            sv = c['saved'][d]
            w = st[b + '/kernel']
            wx, wh, wp = w[:din], w[din:], st[b + '/projection/kernel']
            src, live = sv['src'], sv['live']
            dstep = dout[:, :, d * P:(d + 1) * P][src, bidx].contiguous()              # [T,B,P] step-major
            dpre_all = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
            dpeep_all = torch.empty((T, B, 3 * H), dtype=torch.float32, device=dev) if self.use_peephole else None
            dm_all = torch.empty((T, B, P), dtype=torch.float32, device=dev)
            dc = torch.zeros((B, H), dtype=torch.float32, device=dev)
            dm = torch.zeros((B, P), dtype=torch.float32, device=dev)
            if d_final is not None and d_final[d] is not None:
                dc, dm = d_final[d][0].contiguous().clone(), d_final[d][1].contiguous().clone()
            zero_h = torch.zeros((B, H), dtype=torch.float32, device=dev)
            for s in reversed(range(T)):
                lv = live[s].unsqueeze(1)
                dm_t = (dstep[s] + dm) * lv                   # gradient w.r.t. this step's projected output (live rows)
                dm_carry = dm * (1.0 - lv)                    # finished rows: the state gradient passes through
                dm_all[s].copy_(dm_t)
                dh_raw = ops.gemm(dm_t, wp, transB=True)                               # [B,H]
                _, dc, _, _ = ops.lstm_cell_bwd(dh_raw, dc, zero_h, sv['gates'][s], sv['craw'][s], sv['cprev'][s],
                                                sv['peep'], live[s], want_dpeep=self.use_peephole,
                                                dpre_out=dpre_all[s], dpeep_out=dpeep_all[s] if dpeep_all is not None else None,
                                                cell_clip=self.cell_clip or 0.0)   # LSTMCell clips with tf.clip_by_value:
                                                                                   # no gradient through a clamped state
                dm = ops.gemm(dpre_all[s], wh, transB=True) + dm_carry                 # d m_prev
            dp2d = dpre_all.view(T * B, 4 * H)
            gk = st.g(b + '/kernel')
            # x in step order for this direction: the same gather the forward applied to x W_x
            xs = x[src, bidx].contiguous().view(T * B, din)
            ops.gemm(xs, dp2d, transA=True, out=gk[:din])
            ops.gemm(sv['mprev'].view(T * B, P), dp2d, transA=True, out=gk[din:])
            ops.colsum(dp2d, out=st.g(b + '/bias'))
            ops.gemm(sv['hraw'].view(T * B, H), dm_all.view(T * B, P), transA=True, out=st.g(b + '/projection/kernel'))
            if self.use_peephole:
                dp = ops.colsum(dpeep_all.view(T * B, 3 * H))
                st.g(b + '/w_i_diag').copy_(dp[:H])
                st.g(b + '/w_f_diag').copy_(dp[H:2 * H])
                st.g(b + '/w_o_diag').copy_(dp[2 * H:])
            if need_dx:
                dxs = ops.gemm(dp2d, wx, transB=True).view(T, B, din)                  # step-major
                dxf = torch.zeros_like(dxs)
                dxf[src, bidx] = dxs                           # back to frame order (dpre of dead steps is zero)
                dx += dxf
        self.grad_event = ops.stream_event()
        self.ctx = None
        return dx

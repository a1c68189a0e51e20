"""Bidirectional LSTM encoder on the HIP recurrence kernels.

Mirror of models/encoders/core/blstm.py:13-121 (class BLSTMEncoder, same
constructor arguments and __call__(inputs, inputs_seq_len, keep_prob,
is_training) -> (outputs, final_state)).  Every lstm_impl the reference offers
('BasicLSTMCell', 'LSTMCell', 'LSTMBlockCell', 'LSTMBlockFusedCell',
'CudnnLSTM'; blstm.py:83-121) computes the same cell; here they all map onto
the one fused kernel ('BasicLSTMCell' = no peephole / no clip, blstm.py:124-190).
num_proj is dropped exactly as the reference drops it unless lstm_impl ==
'LSTMCell' (blstm.py:49-52); with 'LSTMCell' it builds tf.contrib.rnn.LSTMCell's
projected cells (blstm.py:187-230) on rnn_util.LSTMPLayer -- the step-by-step fp32
form, for the plain 'blstm' / 'lstm' encoders (the VGG / CLDNN / multitask
front-ends over projected cells raise ValueError: no reference recipe builds them).
"""
import collections
import os as _os

import numpy as np
import torch

from .... import ops
from ...._lib import ASR_F32
from ....utils.parameter import ParamStore
from . import rnn_util
from .rnn_util import LSTMLayer, LSTMPLayer, declare_lstm_vars, declare_lstmp_vars

WARM_BPTT = _os.environ.get('ASR_WARM_BPTT', '1') != '0'    # A-B switch of the read pass ahead of each BPTT kernel
# Two half-batch pipelines (round 6; built, measured, OFF by default).  A recurrence kernel occupies 16 CUs per
# (16-utterance tile, direction) -- 64 - 128 of the 256 at B = 32 / 64 -- and the chip-wide products between two recurrences
# (the next layer's x W_x: 0.5 - 1.1 ms at 4 - 5 x 512, the dx product of the backward pass: 0.4 - 0.9 ms) run strictly
# between them.  With ASR_ENC_HALVES=1 (or encoder.halves = True) a batch of at least two tiles is cut at a tile boundary
# and the two parts go through the layer stack as independent pipelines on two streams (main and lane PIPE_LANE, each with
# its own handle, i.e. its own exchange areas), so that the products of one part run under the recurrences of the other; the
# parts meet again in the outputs (one concatenation) and in the weight gradients (the second part's GEMMs accumulate onto
# the first's: same lanes, fixed order).  Correct (CPU + GPU parity tests run it) and SLOWER (profiles/r06_halves_ab.md): a
# recurrence launch takes as long for half the tiles as for all of them (the chain is T, not B), so a pipeline saves only
# HALF of each product, and two recurrences + a GEMM side by side cost each other more than that: cfg C 58.1 -> 63.8 ms,
# cfg D 71.9 -> 74.6, cfg E 42.2 -> 44.5 with 8 hardware queues; with the runtime's default 4 the fifth stream shares a
# queue and the pipelines serialise (70.6 / 81.1 / 47.8).  (The dropout masks of the two forms differ: a mask is indexed
# by the element's position in ITS part's tensor.)
ENC_HALVES = _os.environ.get('ASR_ENC_HALVES', '0') == '1'
PIPE_LANE = 6

# tf.contrib.rnn.LSTMStateTuple: what the reference's encoders hand back as final state (.c, .h)
LSTMStateTuple = collections.namedtuple('LSTMStateTuple', ('c', 'h'))

LSTM_IMPLS = ('BasicLSTMCell', 'LSTMCell', 'LSTMBlockCell', 'LSTMBlockFusedCell', 'CudnnLSTM')


class _RecurrentEncoderBase(object):
    ndir = 2
    scope_fmt = 'blstm_hidden%d'

    def __init__(self, num_units, num_proj, num_layers, lstm_impl, use_peephole, parameter_init,
                 clip_activation, time_major=True, name='lstm_encoder', dtype=ASR_F32, seed=0):
        assert num_proj != 0
        if lstm_impl not in LSTM_IMPLS:
            raise IndexError('lstm_impl is "BasicLSTMCell" or "LSTMCell" or "LSTMBlockCell" or '
                             '"LSTMBlockFusedCell" or "CudnnLSTM".')
        self.num_units = num_units
        self.num_proj = num_proj if lstm_impl == 'LSTMCell' else None
        if self.num_proj is not None and type(self).__name__ not in ('BLSTMEncoder', 'LSTMEncoder', 'VGGBLSTMEncoder',
                                                                     'VGGLSTMEncoder', 'MultitaskBLSTMEncoder',
                                                                     'MultitaskLSTMEncoder', 'CLDNNEncoder'):
            raise ValueError('LSTMCell projection layers (num_proj) are implemented for the blstm / lstm / vgg_* / '
                             'multitask_* / cldnn_wang encoders, not for %s' % type(self).__name__)
        self.num_layers = num_layers
        self.lstm_impl = lstm_impl
        self.use_peephole = bool(use_peephole) and lstm_impl != 'BasicLSTMCell'
        self.parameter_init = parameter_init
        self.clip_activation = None if lstm_impl == 'BasicLSTMCell' else clip_activation
        self.time_major = time_major
        self.name = name
        self.dtype = ops.dtype_id(dtype)
        self.layer_dtype = self.dtype                # what the caller asked for: the projected layers' operand dtype
        if self.num_proj is not None:
            self.dtype = ASR_F32                     # the projected layers take and return fp32 (rnn_util.LSTMPLayer)
        self.seed = seed
        self.layers = None
        self.store = None
        self.scope_prefix = ''
        self.num_layers_sub = None     # multitask encoders: the layer whose output also feeds the sub-task head
        self.grad_ready_hook = None    # callable(layer_index, LSTMLayer) fired from backward() per finished layer
        self.side_extra = None         # callable run on the side stream after the weight images (once per call)
        self.side_extra_event = None
        self.want_f32_outputs = True   # False: __call__ returns the operand-dtype outputs (no fp32 copy is made)
        self.halves = ENC_HALVES       # two half-batch pipelines where the batch has at least two tiles (see ENC_HALVES)
        self.layers_b = None           # twin layer objects (same variables, own tape) of the second pipeline
        self._split = 0                # rows of the first part in the last __call__ (0: one pipeline)

    # variables are created at graph-build time in the reference; here when the input size is known
    def build(self, store, input_dim, rng, scope_prefix=''):
        self.store = store
        self.scope_prefix = scope_prefix
        self.layers = []
        din = input_dim
        H = self.num_units
        if self.num_proj is not None:
            P = int(self.num_proj)
            for i in range(1, self.num_layers + 1):
                bases = self._declare_projected(store, i, din, P, rng)
                self.layers.append(LSTMPLayer(store, bases, din, H, P, self.use_peephole, 1.0, self.clip_activation,
                                              dtype=self.layer_dtype))
                din = self.ndir * P
            self.output_dim = self.ndir * P
            return self.output_dim
        for i in range(1, self.num_layers + 1):
            bases = self._declare(store, i, din, rng)
            self.layers.append(LSTMLayer(store, bases, din, H, self.use_peephole, 1.0,
                                         self.clip_activation))
            din = self.ndir * H
        self.output_dim = self.ndir * H
        return self.output_dim

    def _declare_projected(self, store, i, din, P, rng):
        scope = self.scope_prefix + (self.scope_fmt % i)
        return declare_lstmp_vars(store, scope, din, self.num_units, P, self.ndir, self.use_peephole,
                                  self.parameter_init, rng)

    def _declare(self, store, i, din, rng):
        scope = self.scope_prefix + (self.scope_fmt % i)
        return declare_lstm_vars(store, scope, din, self.num_units, self.ndir, self.use_peephole,
                                 self.parameter_init, rng)

    def _ensure_built(self, inputs):
        if self.layers is None:
            store = ParamStore(inputs.device)
            self.build(store, inputs.shape[-1], np.random.RandomState(self.seed))
            store.finalize()

    def __call__(self, inputs, inputs_seq_len, keep_prob, is_training, drop_masks=None, rng_state=None):
        """inputs [B,T,input_size] fp32 (cuda); inputs_seq_len [B] int32.
        Returns outputs [T,B,ndir*H] if time_major else [B,T,ndir*H] (fp32) and final_state of
        the last layer: ((c_fw,h_fw),(c_bw,h_bw)) for the bidirectional encoder."""
        self._ensure_built(inputs)
        if self.num_proj is not None:
            return self._call_projected(inputs, inputs_seq_len, keep_prob, is_training, drop_masks, rng_state)
        B, T, _ = inputs.shape
        self.pad_b = (-B) % 16
        if self.pad_b:  # the kernel tiles 16 utterances; pad with zero-length rows
            inputs = torch.cat([inputs, inputs.new_zeros(self.pad_b, T, inputs.shape[2])], 0)
            inputs_seq_len = torch.cat([inputs_seq_len, inputs_seq_len.new_zeros(self.pad_b)], 0)
        self.batch = B
        seq_len = inputs_seq_len.to(torch.int32).contiguous()
        Tt, Bp = T, inputs.shape[0]
        D = inputs.shape[2]
        # the lean GEMM wants a reduction width that is a multiple of its k tile: pad the first layer's rows
        ldk0 = (D + 63) // 64 * 64 if self.dtype != ASR_F32 else D

        if rng_state is None and drop_masks is None and is_training and keep_prob is not None and keep_prob < 1.0:
            # an encoder used on its own (the models pass their own state): a fresh dropout stream per call
            self._dropout_calls = getattr(self, '_dropout_calls', 0) + 1
            rng_state = (self.seed, self._dropout_calls << 40)

        def rs_of(li):
            return (rng_state[0], rng_state[1] + li * (1 << 32)) if rng_state is not None else None

        # side stream, in this order: the weight images of every layer (one launch and one event each: the first
        # layer's GEMM waits for ~10 us of side work, not for the masks), whatever else the model wants refreshed once
        # per step (side_extra: the operand-dtype shadow of the head weights), then the dropout masks (one event each)
        preps, ready_w, ready_m = [], [], []
        with ops.side_lane(inputs.device):
            for li, layer in enumerate(self.layers):
                preps.append(layer.prepare(inputs.device, self.dtype, Tt, Bp, 1.0, False, None, None,
                                           ldk=ldk0 if li == 0 else None))
                ready_w.append(ops.stream_event())
            self.side_extra_event = None
            if self.side_extra is not None:
                self.side_extra()
                self.side_extra_event = ops.stream_event()   # the caller waits for it before using what it refreshed
            for li, layer in enumerate(self.layers):
                dm = drop_masks[li] if drop_masks is not None else None
                preps[li]['mask'] = layer.make_mask(inputs.device, Tt, Bp, keep_prob, is_training, rs_of(li), dm)
                ready_m.append(ops.stream_event())
        self._split = self._split_rows(Bp) if drop_masks is None else 0
        if self._split:
            x, finals = self._forward_halves(inputs, seq_len, preps, ready_w, ready_m, keep_prob, is_training, ldk0)
            return self._finish_call(x, finals, seq_len, B)
        x = ops.bt_to_tb(inputs.contiguous(), self.dtype, ld=ldk0)       # blstm.py:277-279
        final = None
        finals = []
        for li, layer in enumerate(self.layers):
            # the first layer waits for its own weight image (~10 us of side work), the second for the last one's --
            # the lane is in order, so that covers the rest: one marker between two recurrence kernels instead of L - 1
            if not rnn_util.FORK_ONCE or li == 0:
                ops.wait_event(ready_w[li])
            elif li == 1:
                ops.wait_event(ready_w[-1])
            x, final = layer.forward(x, seq_len, self.dtype, keep_prob, is_training, prep=preps[li],
                                     mask_event=ready_m[li])
            finals.append(final)
            if self.num_layers_sub is not None and li + 1 == self.num_layers_sub:
                # blstm.py:326-328: outputs_sub IS the tensor the next layer consumes (after the dropout wrapper)
                self._out_sub_op, self._final_sub_ch = x, final
        return self._finish_call(x, finals, seq_len, B)

    def _finish_call(self, x, finals, seq_len, B):
        self.seq_len_padded = seq_len
        self._out_op = x   # time-major padded-batch outputs in the MFMA operand dtype
        # the fp32 copy the reference's callers receive; models that only consume the operand copy switch it off
        out = ops.cast_to_f32(x) if (x.dtype != torch.float32 and self.want_f32_outputs) else x
        self._out_tm = out
        cf, hf = finals[-1]
        self._final_ch = (cf, hf)          # [ndir,Bp,H] each (the attention bridge consumes these)
        self._finals = finals
        final_state = self._state_tuple(finals, len(finals))
        out_user = out[:, :B]
        if not self.time_major:
            out_user = out_user.transpose(0, 1)
        return out_user, final_state

    # ---- two half-batch pipelines (ENC_HALVES)
    def _split_rows(self, Bp):
        """Rows of the first part (a multiple of 16), or 0 for one pipeline."""
        tiles = Bp // 16
        if not self.halves or tiles < 2 or self.num_proj is not None or self.num_layers_sub is not None:
            return 0
        return ((tiles + 1) // 2) * 16

    def _twins(self):
        if self.layers_b is None:
            self.layers_b = [LSTMLayer(l.store, l.bases, l.din, l.H, l.use_peephole, l.forget_bias, l.cell_clip)
                             for l in self.layers]
        return self.layers_b

    @staticmethod
    def _second_mask(mask):
        """The dropout descriptor of the second part: the same stream, entered 2^30 counters further on (a part's own
        counters stay below 2^29: T B W / 4 with T B W < 2^31, asserted by the recurrence kernels)."""
        return (mask[0], mask[1], mask[2] + (1 << 30)) if isinstance(mask, tuple) else mask

    def _forward_halves(self, inputs, seq_len, preps, ready_w, ready_m, keep_prob, is_training, ldk0):
        dev = inputs.device
        B0, Bp = self._split, inputs.shape[0]
        parts = ((0, B0, self.layers), (B0, Bp, self._twins()))
        start = ops.stream_event()                  # the inputs exist on the main stream here
        xs, fins = [None, None], [[], []]
        sl = [seq_len[lo:hi].contiguous() for lo, hi, _ in parts]
        for li in range(len(self.layers)):
            for h, (lo, hi, layers) in enumerate(parts):
                ctx = ops.side_lane(dev, keep=(inputs, seq_len, sl[1]), lane=PIPE_LANE, after=start) if h else _NoLane()
                with ctx:
                    if li == 0:
                        xs[h] = ops.bt_to_tb(inputs[lo:hi].contiguous(), self.dtype, ld=ldk0)
                        # the second pipeline waits once, for the weight images of every layer (the side lane that builds
                        # them is in order); the first one as the single pipeline does
                        ops.wait_event(ready_w[-1] if h else ready_w[0])
                    elif not h:
                        if not rnn_util.FORK_ONCE:
                            ops.wait_event(ready_w[li])
                        elif li == 1:
                            ops.wait_event(ready_w[-1])
                    prep = preps[li] if not h else dict(preps[li], mask=self._second_mask(preps[li]['mask']))
                    xs[h], fin = layers[li].forward(xs[h], sl[h], self.dtype, keep_prob, is_training, prep=prep,
                                                    mask_event=ready_m[li])
                    fins[h].append(fin)
                    if h and li + 1 == len(self.layers):
                        done = ops.stream_event()
        ops.wait_event(done)
        ops.keep_on_lane(dev, PIPE_LANE, (xs[1],) + tuple(t for f in fins[1] for t in f))
        x = torch.cat([xs[0], xs[1]], 1)
        # final states: the last layer's for the bidirectional stack, every layer's for the unidirectional one
        finals = []
        for li, (a, b) in enumerate(zip(fins[0], fins[1])):
            need = self.ndir == 1 or li + 1 == len(self.layers)
            finals.append((torch.cat([a[0], b[0]], 1), torch.cat([a[1], b[1]], 1)) if need else (None, None))
        return x, finals

    def _call_projected(self, inputs, inputs_seq_len, keep_prob, is_training, drop_masks, rng_state):
        """lstm_impl='LSTMCell' with num_proj: the stack on LSTMPLayer.  Outputs [T,B,ndir*P]; final state
        ((c_fw [B,H], m_fw [B,P]), (c_bw, m_bw)) of the last layer (one tuple per layer for the unidirectional stack)."""
        B, T, _ = inputs.shape
        self.pad_b, self.batch = 0, B
        seq_len = inputs_seq_len.to(torch.int32).contiguous()
        x = ops.bt_to_tb(inputs.contiguous(), ASR_F32)
        if rng_state is None and drop_masks is None and is_training and keep_prob is not None and keep_prob < 1.0:
            self._dropout_calls = getattr(self, '_dropout_calls', 0) + 1
            rng_state = (self.seed, self._dropout_calls << 40)
        finals = []
        for li, layer in enumerate(self.layers):
            mask = None
            if is_training and drop_masks is not None:
                mask = drop_masks[li]
            elif is_training and rng_state is not None and keep_prob < 1.0:
                mask = ops.dropout_mask((T, B, self.ndir * layer.P), keep_prob, rng_state[0],
                                        rng_state[1] + li * (1 << 32), x.device)
            x, fin = layer.forward(x, seq_len, mask)
            finals.append(fin)
            if self.num_layers_sub is not None and li + 1 == self.num_layers_sub:
                self._out_sub_op, self._final_sub_ch = x, fin
        self.seq_len_padded = seq_len
        self._out_op = self._out_tm = x
        self._finals = finals
        self._final_ch = None
        out_user = x if self.time_major else x.transpose(0, 1)
        return out_user, self._state_tuple_projected(finals, len(finals))

    def _state_tuple_projected(self, finals, upto):
        """_state_tuple for the projected stack: finals[li] = per direction (c [B,H], m [B,P])."""
        if self.ndir == 2:
            return tuple(LSTMStateTuple(c, m) for c, m in finals[upto - 1])
        return tuple(LSTMStateTuple(*f[0]) for f in finals[:upto])

    def _state_tuple(self, finals, upto):
        """The reference's final_state: bidirectional_dynamic_rnn of the LAST layer -> (LSTMStateTuple fw,
        LSTMStateTuple bw) (blstm.py:313-323); MultiRNNCell under one dynamic_rnn -> one LSTMStateTuple per layer
        (lstm.py:275-284).  `upto`: number of layers the (sub-)stack has."""
        B = self.batch
        if self.ndir == 2:
            cf, hf = finals[upto - 1]
            return tuple(LSTMStateTuple(cf[d, :B], hf[d, :B]) for d in range(2))
        return tuple(LSTMStateTuple(cf[0, :B], hf[0, :B]) for cf, hf in finals[:upto])

    def backward(self, d_outputs, d_final=None, need_input_grad=False, d_outputs_sub=None):
        """d_outputs: gradient w.r.t. the TIME-MAJOR padded-batch outputs [T,Bpad,ndir*H] fp32.
        d_outputs_sub (multitask encoders): gradient w.r.t. the sub-task outputs, joined where they branch off.
        Returns the gradient w.r.t. the (time-major) encoder input if need_input_grad."""
        if self.num_proj is not None:
            dx = d_outputs
            for li in reversed(range(len(self.layers))):
                if d_outputs_sub is not None and li == self.num_layers_sub - 1:
                    dx = torch.add(dx, d_outputs_sub)            # the sub-task head's gradient joins where it branched off
                dx = self.layers[li].backward(dx.contiguous(), None, need_dx=(li > 0 or need_input_grad))
                if self.grad_ready_hook is not None:
                    self.grad_ready_hook(li, self.layers[li])
            ops.join_side(d_outputs.device)
            return dx
        if self._split:
            return self._backward_halves(d_outputs, d_final, need_input_grad)
        dx = d_outputs
        masked = False
        for li in reversed(range(len(self.layers))):
            if d_outputs_sub is not None and li == self.num_layers_sub - 1:
                dx = torch.add(dx, d_outputs_sub)
            dcf = dhf = None
            if d_final is not None and li == len(self.layers) - 1:
                dcf, dhf = d_final
            # the dropout mask of the layer below is folded into this layer's dx GEMM (no separate pass over dx
            # between two BPTT kernels) -- unless another gradient joins dx before that layer's mask applies
            below = self.layers[li - 1].ctx['mask'] if li > 0 else None
            if d_outputs_sub is not None and li - 1 == self.num_layers_sub - 1:
                below = None
            # read pass over the saved activations of the layer below while this layer's dx product runs -- when they
            # fit the 256 MB memory-side cache with room to spare (76 MB at cfg B; 306 MB at 5x512 / B = 32 do not)
            cb = self.layers[li - 1].ctx if (li > 0 and WARM_BPTT) else None
            if cb is not None and (cb['gates'].numel() * cb['gates'].element_size() +
                                   cb['cs'].numel() * cb['cs'].element_size()) > (128 << 20):
                cb = None
            dx = self.layers[li].backward(dx.contiguous(), dcf, dhf, need_dx=(li > 0 or need_input_grad),
                                          dout_masked=masked, dx_mask=below, background=(li > 0),
                                          warm=(cb['gates'], cb['cs']) if cb is not None else None)
            ops.wait_event(getattr(self.layers[li], 'warm_event', None))
            masked = below is not None
            if self.grad_ready_hook is not None:      # data-parallel step: this layer's gradients are on their way
                self.grad_ready_hook(li, self.layers[li])
        ops.join_side(d_outputs.device)      # weight-gradient GEMMs issued on the side stream
        return dx


    def _backward_halves(self, d_outputs, d_final, need_input_grad):
        """backward() of a call that went through _forward_halves: the same two pipelines, top layer first; per layer the
        first part leaves its weight-gradient products to the second, whose twin layer writes the variables."""
        dev = d_outputs.device
        B0 = self._split
        dxs = [d_outputs[:, :B0].contiguous(), d_outputs[:, B0:].contiguous()]
        dfin = [(None, None), (None, None)]
        if d_final is not None:
            dcf, dhf = d_final
            dfin = [(dcf[:, :B0].contiguous(), dhf[:, :B0].contiguous()), (dcf[:, B0:].contiguous(), dhf[:, B0:].contiguous())]
        start = ops.stream_event()
        stacks = (self.layers, self.layers_b)
        masked = [False, False]
        top = len(self.layers) - 1
        for li in reversed(range(len(self.layers))):
            acc = None
            for h, layers in enumerate(stacks):
                ctx = ops.side_lane(dev, keep=(dxs[1],) + dfin[1], lane=PIPE_LANE, after=start) if h else _NoLane()
                with ctx:
                    below = layers[li - 1].ctx['mask'] if li > 0 else None
                    cb = layers[li - 1].ctx if (li > 0 and WARM_BPTT) else None
                    if cb is not None and (cb['gates'].numel() * cb['gates'].element_size() +
                                           cb['cs'].numel() * cb['cs'].element_size()) > (64 << 20):
                        cb = None
                    dcf, dhf = dfin[h] if li == top else (None, None)
                    dxs[h] = layers[li].backward(dxs[h].contiguous(), dcf, dhf, need_dx=(li > 0 or need_input_grad),
                                                 dout_masked=masked[h], dx_mask=below, background=(li > 0),
                                                 warm=(cb['gates'], cb['cs']) if cb is not None else None,
                                                 acc=acc, finish=bool(h))
                    ops.wait_event(getattr(layers[li], 'warm_event', None))
                    masked[h] = below is not None
                    if not h:
                        acc = layers[li].acc
                        layers[li].acc = None
                    elif li == 0:
                        done = ops.stream_event()
            if self.grad_ready_hook is not None:      # the twin finished the layer: its event covers both parts
                self.grad_ready_hook(li, self.layers_b[li])
        ops.wait_event(done)
        ops.join_side(dev)
        if not need_input_grad:
            return None
        return torch.cat([dxs[0], dxs[1]], 1)


class _NoLane(object):
    """The main stream as a context (the first pipeline runs where the caller stands)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class BLSTMEncoder(_RecurrentEncoderBase):
    """models/encoders/core/blstm.py:13 BLSTMEncoder."""
    ndir = 2
    scope_fmt = 'blstm_hidden%d'

"""CLDNN (CNN + bidirectional LSTM + DNN) encoder -- mirror of models/encoders/core/cldnn_wang.py:46-267
(class CLDNNEncoder, after Wang et al., arXiv:1702.07793), same constructor arguments as VGGBLSTMEncoder.

__call__(inputs [B,T,num_channels*(splice*num_stack)*3], inputs_seq_len, keep_prob, is_training):
reshape to [B*T, num_channels, splice*num_stack, 3] (:134-136); CNN1 conv 11x21 stride (3,2) 3->32 + relu, CNN2 conv
11x11 stride (1,2) 32->32 + relu, CNN3 conv 3x3 32->96 + relu, each followed by a 1x1 / stride-1 max_pool (the
identity) and dropout (:139-177); flatten; the BLSTM stack of blstm.py (:185-220); fc1 896 relu + dropout, fc2 74 relu
(:227-249) -> outputs [T,B,74].  Variables: CNN{1,2,3}/conv/{weight,bias} (tf.Variable in cnn_util.py:66-69),
blstm_hidden<i>/..., fc{1,2}/{weights,biases}.

Execution: every convolution is asr_im2col (any kernel / stride, TensorFlow's SAME padding) + MFMA GEMM with fused
bias + ReLU, chunked over frames so that the patch matrix stays a bounded scratch (K = 3872 for CNN2); the backward
pass recomputes the patches.  All B*T frames are convolved, as in the reference: the recurrent stack masks what lies
beyond an utterance, so nothing of a padded frame reaches a gradient.
"""
import numpy as np
import torch

from .... import ops
from ...._lib import ASR_BF16, ASR_F32
from ....utils.parameter import ParamStore
from .blstm import BLSTMEncoder
from .vgg_blstm import _trunc_normal

# (scope, kernel (kh, kw), stride (sh, sw), Cin, Cout)
CONVS = [('CNN1/conv', (11, 21), (3, 2), 3, 32), ('CNN2/conv', (11, 11), (1, 2), 32, 32),
         ('CNN3/conv', (3, 3), (1, 1), 32, 96)]
CHUNK_FRAMES = 1024
FC1, FC2 = 896, 74


class _ConvStack(object):
    def __init__(self, input_size, splice, num_stack, parameter_init, dtype):
        assert input_size % 3 == 0
        self.F = input_size // 3
        self.W = splice * num_stack
        self.parameter_init = parameter_init
        self.dtype = dtype
        self.prefix = ''
        h, w = self.F, self.W
        self.shapes = []
        for _, (kh, kw), (sh, sw), cin, cout in CONVS:
            self.shapes.append((h, w, cin))
            h, w = ops.conv_out_hw(h, w, sh, sw)
        self.out_hw = (h, w)
        self.out_dim = h * w * CONVS[-1][4]

    def build(self, store, rng, prefix=''):
        self.store, self.prefix = store, prefix
        for name, (kh, kw), _, cin, cout in CONVS:
            store.declare(prefix + name + '/weight', (kh, kw, cin, cout),
                          _trunc_normal(rng, self.parameter_init, (kh, kw, cin, cout)))
            store.declare(prefix + name + '/bias', (cout,), np.zeros(cout))

    def _patches(self, xc, conv):
        _, (kh, kw), (sh, sw), cin, _ = conv
        K = kh * kw * cin
        return ops.im2col(xc, kh, kw, sh, sw, ldp=(K + 7) // 8 * 8)[:, :K]

    def forward(self, x_btd, keep_prob, is_training, rng_state):
        """x [B,T,F*W*3] fp32 cuda -> [B,T,out_dim] fp32; keeps what backward needs."""
        st = self.store
        sh_ = st.shadow(self.dtype)
        B, T, Dd = x_btd.shape
        assert Dd == self.F * self.W * 3
        N = B * T
        x = x_btd.reshape(N, self.F, self.W, 3).contiguous()
        if self.dtype == ASR_BF16:
            x = ops.cast_from_f32(x, self.dtype)
        drop = is_training and keep_prob < 1.0
        acts, masks, relu_outs = [x], [], []
        for li, conv in enumerate(CONVS):
            name, (kh, kw), (s_h, s_w), cin, cout = conv
            Hi, Wi, _ = self.shapes[li]
            Ho, Wo = ops.conv_out_hw(Hi, Wi, s_h, s_w)
            w2d = sh_[self.prefix + name + '/weight'].view(kh * kw * cin, cout)
            out = torch.empty((N, Ho, Wo, cout), dtype=x.dtype, device=x.device)
            for c0 in range(0, N, CHUNK_FRAMES):
                xc = acts[-1][c0:c0 + CHUNK_FRAMES]
                y = ops.gemm(self._patches(xc, conv), w2d, bias=st[self.prefix + name + '/bias'], relu=True)
                out[c0:c0 + CHUNK_FRAMES] = y.view(xc.shape[0], Ho, Wo, cout)
            mask = None
            fed = out
            if drop:      # tf.nn.dropout after the (identity) max_pool of each block
                seed, off = rng_state
                mask = ops.dropout_mask(out.shape, keep_prob, seed + 13, off + ((li + 1) << 32), out.device)
                fed = ops.apply_mask(out, mask)
            masks.append(mask)
            acts.append(fed)
            relu_outs.append(out)                 # the ReLU output the backward pass gates on
        self.ctx = dict(B=B, T=T, N=N, acts=acts, masks=masks, relu_outs=relu_outs)
        y = acts[-1].reshape(N, self.out_dim)
        y = ops.cast_to_f32(y) if y.dtype != torch.float32 else y
        return y.view(B, T, self.out_dim)

    def backward(self, dout_btd):
        """dout [B,T,out_dim] fp32 (gradient w.r.t. the flattened CNN3 output)."""
        c, st = self.ctx, self.store
        sh_ = st.shadow(self.dtype)
        N = c['N']
        Ho, Wo = self.out_hw
        d = dout_btd.reshape(N, Ho, Wo, CONVS[-1][4]).contiguous()
        for li in reversed(range(len(CONVS))):
            name, (kh, kw), (s_h, s_w), cin, cout = CONVS[li]
            Hi, Wi, _ = self.shapes[li]
            oh, ow = ops.conv_out_hw(Hi, Wi, s_h, s_w)
            x_in = c['acts'][li]
            out = c['relu_outs'][li]
            mask = c['masks'][li]
            w2d = sh_[self.prefix + name + '/weight'].view(kh * kw * cin, cout)
            gw = st.g(self.prefix + name + '/weight').view(kh * kw * cin, cout)
            gb = st.g(self.prefix + name + '/bias')
            gb_acc = torch.zeros_like(gb)
            need_dx = li > 0
            din = torch.empty((N, Hi, Wi, cin), dtype=torch.float32, device=d.device) if need_dx else None
            for ci, c0 in enumerate(range(0, N, CHUNK_FRAMES)):
                sl = slice(c0, c0 + CHUNK_FRAMES)
                n = out[sl].shape[0]
                dpre = ops.relu_bwd(d[sl].contiguous(), out[sl].contiguous(),
                                    mask[sl].contiguous() if mask is not None else None).view(n * oh * ow, cout)
                ops.gemm(self._patches(x_in[sl].contiguous(), CONVS[li]), dpre, transA=True, out=gw, accumulate=(ci > 0))
                gb_acc += ops.colsum(dpre)
                if need_dx:
                    dpat = ops.gemm(dpre, w2d, transB=True, out_dtype=ASR_F32)
                    din[sl] = ops.col2im(dpat, n, Hi, Wi, cin, kh, kw, s_h, s_w)
            gb.copy_(gb_acc)
            d = din
        self.ctx = None


class CLDNNEncoder(BLSTMEncoder):
    """models/encoders/core/cldnn_wang.py:46 CLDNNEncoder."""

    def __init__(self, input_size, splice, num_stack, num_units, num_proj, num_layers, lstm_impl, use_peephole,
                 parameter_init, clip_activation, time_major=False, name='cldnn_wang_encoder', dtype=ASR_F32, seed=0):
        BLSTMEncoder.__init__(self, num_units, num_proj, num_layers, lstm_impl, use_peephole, parameter_init,
                              clip_activation, time_major, name, dtype, seed)
        self.front = _ConvStack(input_size, splice, num_stack, parameter_init, ops.dtype_id(dtype))
        self.input_size, self.splice, self.num_stack = input_size, splice, num_stack

    def build(self, store, input_dim, rng, scope_prefix=''):
        assert input_dim == self.front.F * self.front.W * 3
        self.front.build(store, rng, scope_prefix)
        E = super(CLDNNEncoder, self).build(store, self.front.out_dim, rng, scope_prefix)
        p = scope_prefix
        store.declare(p + 'fc1/weights', (E, FC1), _trunc_normal(rng, self.parameter_init, (E, FC1)))
        store.declare(p + 'fc1/biases', (FC1,), np.zeros(FC1))
        store.declare(p + 'fc2/weights', (FC1, FC2), _trunc_normal(rng, self.parameter_init, (FC1, FC2)))
        store.declare(p + 'fc2/biases', (FC2,), np.zeros(FC2))
        self._p = p
        self.output_dim = FC2
        return FC2

    def __call__(self, inputs, inputs_seq_len, keep_prob, is_training, drop_masks=None, rng_state=None):
        if self.layers is None:
            store = ParamStore(inputs.device)
            self.build(store, inputs.shape[-1], np.random.RandomState(self.seed))
            store.finalize()
        st = self.store
        rs = rng_state
        if rs is None:
            # an encoder used on its own (the models pass their state): fresh masks for the conv stack / fc1 every
            # training call, as tf.nn.dropout draws them (own seed: the BLSTM stack below counts its calls itself)
            if is_training and keep_prob is not None and float(keep_prob) < 1.0:
                self._front_calls = getattr(self, '_front_calls', 0) + 1
            rs = (self.seed + 7, getattr(self, '_front_calls', 0) << 40)
        x = self.front.forward(inputs.contiguous(), float(keep_prob), is_training, rs)
        want = self.want_f32_outputs
        self.want_f32_outputs = False
        _, final_state = super(CLDNNEncoder, self).__call__(x, inputs_seq_len, keep_prob, is_training, drop_masks,
                                                            rng_state)
        self.want_f32_outputs = want
        h = self._out_op                                    # [T,Bp,2H] in the operand dtype
        T, Bp, E = h.shape
        sh_ = st.shadow(self.dtype)
        a1 = ops.gemm(h.view(T * Bp, E), sh_[self._p + 'fc1/weights'], bias=st[self._p + 'fc1/biases'], relu=True)
        m1, a1d = None, a1
        if is_training and keep_prob < 1.0:
            m1 = ops.dropout_mask(a1.shape, keep_prob, rs[0] + 17, rs[1] + (9 << 32), a1.device)
            a1d = ops.apply_mask(a1, m1)
        a2 = ops.gemm(a1d, sh_[self._p + 'fc2/weights'], bias=st[self._p + 'fc2/biases'], relu=True)
        self._dnn = dict(h=h, a1=a1, a1d=a1d, m1=m1, a2=a2)
        self._out_op = a2.view(T, Bp, FC2)
        out = ops.cast_to_f32(self._out_op) if (a2.dtype != torch.float32 and want) else self._out_op
        self._out_tm = out
        out_user = out[:, :self.batch]
        if not self.time_major:
            out_user = out_user.transpose(0, 1)
        return out_user, final_state

    def backward(self, d_outputs, d_final=None, need_input_grad=False, d_outputs_sub=None):
        """d_outputs [T,Bpad,74] fp32."""
        st, c = self.store, self._dnn
        sh_ = st.shadow(self.dtype)
        T, Bp, _ = d_outputs.shape
        E = c['h'].shape[2]
        d2 = ops.relu_bwd(d_outputs.reshape(T * Bp, FC2).contiguous(), c['a2'], None)
        ops.gemm(c['a1d'], d2, transA=True, out=st.g(self._p + 'fc2/weights'))
        ops.colsum(d2, out=st.g(self._p + 'fc2/biases'))
        da1 = ops.gemm(d2, sh_[self._p + 'fc2/weights'], transB=True, out_dtype=ASR_F32)
        d1 = ops.relu_bwd(da1, c['a1'], c['m1'])
        ops.gemm(c['h'].view(T * Bp, E), d1, transA=True, out=st.g(self._p + 'fc1/weights'))
        ops.colsum(d1, out=st.g(self._p + 'fc1/biases'))
        dh = ops.gemm(d1, sh_[self._p + 'fc1/weights'], transB=True, out_dtype=ASR_F32)
        self._dnn = None
        dx = super(CLDNNEncoder, self).backward(dh.view(T, Bp, E), d_final, need_input_grad=True)   # [T,Bp,out_dim]
        B = self.batch
        self.front.backward(dx[:, :B].transpose(0, 1).contiguous())
        return None
